#!/bin/bash
# device-resident rates of other ER shapes on the same kernels (tier cliffs): bench.py --workload er_<n>_<cols>_<m>_<k>
# usage: tools/sweep_er_shapes.sh [shape ...]   -> one line per shape
SHAPES=${@:-"er_500000_20000000_500000_8 er_200000_16000000_200000_8 er_200000_16000000_200000_6 er_1000000_20000000_1000000_14 er_1000000_20000000_1000000_12 er_1000000_20000000_1000000_8"}
for w in $SHAPES; do
  timeout -k 10 300 python bench.py --workload $w --steps 4 --warmup 1 --no-extras --no-cpu-baseline 2> gpurun_out/sweep.err > gpurun_out/sweep.json || { echo "$w FAILED"; tail -2 gpurun_out/sweep.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/sweep.json')); r=d['roofline']
print('$w', round(d['value']/1e6,2), 'M/s  ms/step', d['ms_per_step'], ' kernel', r['kernel'], 'grid', r['grid'], 'lds', r['lds_bytes_per_block'], 'walk ms', r['kernel_ms'], 'frac', r['frac'])"
done
