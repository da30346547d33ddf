#!/bin/bash
# AddressSanitizer + UBSan run of the HOST side of the library on the CPU (GPU sanitizers are not available on the pool):
# builds ugs_host.cpp / ugs_apx.cpp instrumented, links them with the normal device objects, and runs the CPU tests that
# drive the host code (preprocessing vs the oracle, the C ABI, the apx entry point) with the runtime preloaded.
set -e
cd "$(dirname "$0")/../ss-gnn_amd/csrc"
python3 build.py > /dev/null
for f in ugs_host.cpp ugs_apx.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -x hip -c $f -o /tmp/asan_${f%.*}.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o /tmp/libugs_asan.so /tmp/asan_ugs_host.o /tmp/asan_ugs_apx.o ugs_kernels.o ugs_eps.o ugs_preproc.o ugs_collate.o ugs_batch.o ugs_apx_gpu.o
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
cd ../..
UGS_MI355_LIB=/tmp/libugs_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 python3 -m pytest tests/test_cabi_and_host.py tests/test_apx_entry.py -x -q
