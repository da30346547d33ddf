#!/bin/bash
# Front-end probe of the walk kernel: A/B of the padded diagnostic builds (tools/mkvar.sh P1..P4 -DUGS_PAD=1..4: 64 dummy
# instructions per growth step -- scalar 4 B / scalar 8 B / vector 4 B / vector 8 B) against the shipped kernel, then one
# counter pass of the instruction-cache and dual-issue counters.  usage (GPU box): tools/pad_probe.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tools/ab.sh base P1 P2 P3 P4 2>&1 | tee gpurun_out/pad_probe.txt
OUT=gpurun_out/pmc_front
mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-extras --no-cpu-baseline"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.json 2> $OUT/$name.err; }
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_BUSY_CYCLES && \
run ic2 SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU GRBM_GUI_ACTIVE
echo "exit $?"; ls $OUT
