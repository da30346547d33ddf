#!/bin/bash
# PMC passes of the bench command (each counter group in its own rocprofv3 run, with --kernel-trace only).
# usage: tools/profile_pmc.sh <tag> [bench args...]     -> gpurun_out/pmc_<tag>/{sq1,sq2,fetch,write}/...
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-extras --no-cpu-baseline $*"
rocprofv3 -L > $OUT/counters_list.txt 2>&1
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.json 2> $OUT/$name.err; }
run fetch FETCH_SIZE && run write WRITE_SIZE && \
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR && \
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
if [ -n "$PMC_MORE" ]; then   # front-end and latency counters (separate passes)
run sq3 SQ_INSTS SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES && \
run sq4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES
fi
echo "exit $?"; ls $OUT
