#!/bin/bash
# usage: tools/ab.sh [-w workload] A B C ...   -> bench with ab/lib_<X>.so alternately, two rounds on the same box (the first round
# also checks 20000 rows against the oracle); prints k-subgraphs/s, walk / fill kernel ms and the parity row count per run
WL=c5_er_1m
if [ "$1" = "-w" ]; then WL=$2; shift 2; fi
mkdir -p gpurun_out/ab
for round in 1 2; do for v in "$@"; do
  if [ $round = 1 ]; then EXTRA="--cpu-sample 20000 --no-cpu-reference"; else EXTRA="--no-cpu-baseline"; fi
  UGS_MI355_LIB=$PWD/ab/lib_$v.so timeout -k 10 300 python bench.py --workload $WL --steps 6 --warmup 2 --no-extras $EXTRA > gpurun_out/ab/$v.$WL.json 2> gpurun_out/ab/$v.$WL.err || { echo "$v FAILED"; tail -3 gpurun_out/ab/$v.$WL.err; continue; }
  python -c "import json; d=json.load(open('gpurun_out/ab/$v.$WL.json')); print('$v', '$WL', round(d['value']/1e6,2), 'M/s walk', d['roofline']['kernel_ms'], 'ms fill', d['roofline']['path']['fill_kernel_ms'], 'parity rows', d.get('parity_checked_rows'))"
done; done
