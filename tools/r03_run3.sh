#!/bin/bash
mkdir -p gpurun_out/r03g
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tools/profile_pmc.sh r03_w --workload er_500000_20000000_500000_8 > gpurun_out/r03g/pmc_w.log 2>&1
tools/profile_pmc.sh r03_v --workload er_200000_16000000_200000_8 > gpurun_out/r03g/pmc_v.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/pmc_r03_w > gpurun_out/r03g/r03_tierW_pmc_summary.json
python3 tools/summarize_pmc.py gpurun_out/pmc_r03_v > gpurun_out/r03g/r03_tierV_pmc_summary.json
timeout -k 10 700 python tools/rank0_emulation.py > gpurun_out/r03g/emul.log 2>&1; echo "emul rc $?"; grep -E "^auto|^best" gpurun_out/r03g/emul.log | cut -c1-700
timeout -k 10 300 python bench.py --gpus 2 --rehearse --workload er_200000_4000000_100000_8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03g/rehearse.json 2> gpurun_out/r03g/rehearse.err; echo "rehearse rc $?"
timeout -k 10 400 python bench.py > gpurun_out/r03g/bench_full.json 2> gpurun_out/r03g/bench_full.err; echo "bench rc $?"
