#!/bin/bash
mkdir -p gpurun_out/r03e
timeout -k 10 300 python -m pytest tests/test_gpu_batch_pass.py tests/test_gpu_apx.py -x -q -m gpu 2>&1 | tail -3
python tools/batch_pass_probe.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python tools/rank0_emulation.py > gpurun_out/r03e/emul.log 2>&1; echo "emul rc $?"; grep -E "^auto|^best" gpurun_out/r03e/emul.log | cut -c1-900
timeout -k 10 300 python bench.py --gpus 2 --rehearse --workload er_200000_4000000_100000_8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03e/rehearse.json 2> gpurun_out/r03e/rehearse.err; echo "rehearse rc $?"; tail -3 gpurun_out/r03e/rehearse.err
timeout -k 10 300 python bench.py --force-collate --no-extras --no-cpu-baseline > gpurun_out/r03e/force_collate.json 2> gpurun_out/r03e/force_collate.err; echo "fc rc $?"; cut -c1-300 gpurun_out/r03e/force_collate.json
timeout -k 10 300 python -m pytest tests/test_gpu_safety.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
