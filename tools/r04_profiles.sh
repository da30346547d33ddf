#!/bin/bash
# round-4 evidence session: kernel stats (rocprofv3 --kernel-trace --stats) of C5 / C3 / C4 as the bench's main workload, PMC passes of
# the walk kernel on C5 (instruction mix, issue ports, HBM bytes: separate passes, --kernel-trace only), stamp profile of the tier-V
# workload (diagnostic build ab/lib_stamps.so: tools/mkvar.sh stamps -DUGS_STAMPS), the full default bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p; mkdir -p $O
for w in c5_er_1m c3_proteins_b8192 c4_qm9_b65536; do
  S=10; [ $w = c5_er_1m ] || S=200
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python3 bench.py --workload $w --steps $S --warmup 3 --no-extras --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || echo "kt $w failed"
  find $O/kt_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${w}_kernel_stats.csv
done
PMC_MORE=1 tools/profile_pmc.sh r04_c5 > $O/pmc_c5.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/pmc_r04_c5 > $O/r04_c5_pmc_summary.json
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_c5_plain.json 2> $O/bench_c5_plain.err
python3 tools/traffic_from_pmc.py $O/r04_c5_pmc_summary.json $O/bench_c5_plain.json > $O/traffic.log 2>&1 && cp profiles/pmc_traffic.json $O/pmc_traffic.json   # copy it back into profiles/ (source path: profiles/r04_c5_pmc_summary.json)
if [ -f ab/lib_stamps.so ]; then
  for w in c5_er_1m er_200000_16000000_200000_8; do STAMPS_LIB=ab/lib_stamps.so timeout -k 10 200 python tools/stamps.py $w > $O/stamps_$w.txt 2>&1; done
fi
timeout -k 10 500 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc $?"
ls $O | head -40
