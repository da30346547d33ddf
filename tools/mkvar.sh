#!/bin/bash
# usage: tools/mkvar.sh <name> [extra hipcc flags...]  -> ab/lib_<name>.so (kernels + host rebuilt with the flags, other objects reused)
# A/B variants of the walk kernel for same-box comparisons (tools/ab.sh); ab/ is git-ignored (*.so) but travels to the GPU box.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
HERE=$ROOT/ss-gnn_amd/csrc
W=/tmp/vb/$NAME; mkdir -p $W $ROOT/ab
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip"
/opt/rocm/bin/hipcc $FL -mllvm -amdgpu-sched-strategy=${SCHED:-max-ilp} "$@" -c $HERE/ugs_kernels.hip -o $W/ugs_kernels.o &
/opt/rocm/bin/hipcc $FL "$@" -c $HERE/ugs_host.cpp -o $W/ugs_host.o &
wait
for f in ugs_eps ugs_preproc ugs_apx ugs_apx_gpu ugs_collate ugs_batch; do [ -f $HERE/$f.o ] || { echo "missing $HERE/$f.o (run build.py)"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab/lib_$NAME.so $W/ugs_kernels.o $W/ugs_host.o $HERE/ugs_eps.o $HERE/ugs_preproc.o $HERE/ugs_apx.o $HERE/ugs_apx_gpu.o $HERE/ugs_collate.o $HERE/ugs_batch.o
echo built ab/lib_$NAME.so
