#!/bin/bash
# round-3 evidence session: kernel stats (rocprofv3 --kernel-trace --stats) of C5 / C3 / C4 as the bench's main workload, PMC passes
# of the 8-lane tier on C3 and C4, the full default bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03p; mkdir -p $O
for w in c5_er_1m c3_proteins_b8192 c4_qm9_b65536; do
  S=10; [ $w = c5_er_1m ] || S=200
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python3 bench.py --workload $w --steps $S --warmup 3 --no-extras --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || echo "kt $w failed"
  find $O/kt_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${w}_kernel_stats.csv
done
PMC_MORE=1 tools/profile_pmc.sh r03_c3 --workload c3_proteins_b8192 > $O/pmc_c3.log 2>&1
PMC_MORE=1 tools/profile_pmc.sh r03_c4 --workload c4_qm9_b65536 > $O/pmc_c4.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/pmc_r03_c3 > $O/r03_c3_pmc_summary.json
python3 tools/summarize_pmc.py gpurun_out/pmc_r03_c4 > $O/r03_c4_pmc_summary.json
timeout -k 10 400 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc $?"
ls $O | head -40
