#!/bin/bash
# one GPU session: multi-GPU emulation + rehearsal, A/B of the shift variant, PC sampling of the walk kernel
mkdir -p gpurun_out/r03b
timeout -k 10 500 python tools/rank0_emulation.py > gpurun_out/r03b/emul.log 2>&1; echo "emul rc $?"
timeout -k 10 300 python bench.py --gpus 2 --rehearse --workload er_200000_4000000_100000_8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03b/rehearse.json 2> gpurun_out/r03b/rehearse.err; echo "rehearse rc $?"
timeout -k 10 300 python bench.py --force-collate --no-extras --no-cpu-baseline > gpurun_out/r03b/force_collate.json 2> gpurun_out/r03b/force_collate.err; echo "fc rc $?"
tools/ab.sh base shift 2>&1 | tee gpurun_out/r03b/ab_shift.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
UGS_MI355_LIB=$PWD/ab/lib_baseg.so timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 1048576 --output-format csv -d gpurun_out/r03b/pcs -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r03b/pcs.json 2> gpurun_out/r03b/pcs.err; echo "pcs rc $?"
ls -la gpurun_out/r03b/pcs/* 2>/dev/null | head; tail -3 gpurun_out/r03b/pcs.err
