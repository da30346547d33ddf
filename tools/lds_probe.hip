// tools/lds_probe.hip -- what one wave-wide LDS operation costs on MI355X when the whole CU is busy with them
// (16 one-wave blocks per CU, 8 KB of LDS each, like the tier-M walk kernel).  Prints CU-cycles per wave instruction for
// plain reads/writes, 16-bit writes and the atomics the walk kernel uses, with sequential and with hashed addresses.
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o gpurun_out/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int WORDS = 2048;       // 8 KB per wave
constexpr int ITERS = 2048;

enum Op { RD32, WR32, WR16, ADD, ADD_RTN, MIN_RTN, CAS_RTN, RD64 };

template <int OP, bool RANDOM>
__global__ __launch_bounds__(64, 4) void probe(unsigned *out, unsigned long long *cycles) {
    __shared__ unsigned lds[WORDS];
    const unsigned lane = threadIdx.x;
    for (int i = lane; i < WORDS; i += 64) lds[i] = i;
    __syncthreads();
    unsigned x = lane * 2654435761u + blockIdx.x, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int it = 0; it < ITERS; ++it) {
        unsigned idx;
        if (RANDOM) { x = x * 1664525u + 1013904223u; idx = (x >> 12) & (WORDS - 1); }
        else idx = (lane + it * 64) & (WORDS - 1);
        if (OP == RD32) acc += *reinterpret_cast<volatile unsigned *>(&lds[idx]);
        else if (OP == RD64) { idx &= ~1u; unsigned long long v = *reinterpret_cast<volatile unsigned long long *>(&lds[idx]); acc += (unsigned)v + (unsigned)(v >> 32); }
        else if (OP == WR32) lds[idx] = it;
        else if (OP == WR16) reinterpret_cast<unsigned short *>(lds)[RANDOM ? (x >> 11) & (2 * WORDS - 1) : (lane + it * 64) & (2 * WORDS - 1)] = (unsigned short)it;
        else if (OP == ADD) atomicAdd(&lds[idx], 1u);
        else if (OP == ADD_RTN) acc += atomicAdd(&lds[idx], 1u);
        else if (OP == MIN_RTN) acc += atomicMin(&lds[idx], (unsigned)it);
        else if (OP == CAS_RTN) acc += atomicCAS(&lds[idx], (unsigned)it, lane);
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = acc + lds[lane];
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP, bool RANDOM>
void run(const char *name, unsigned *out, unsigned long long *cyc, int blocks, int cus) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<OP, RANDOM><<<blocks, 64>>>(out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<OP, RANDOM><<<blocks, 64>>>(out, cyc);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // wave instructions per CU = blocks/cus * ITERS; CU cycles = ms * clock
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double cu_cycles = ms * 1e-3 * khz * 1e3;
    const double per_inst = cu_cycles / ((double)blocks / cus * ITERS);
    printf("%-10s %-10s %8.3f ms  %6.2f CU-cycles per wave instruction (incl. address arithmetic)\n", name, RANDOM ? "hashed" : "sequential", ms, per_inst);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 16;
    unsigned *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * 64 * 4); hipMalloc(&cyc, (size_t)blocks * 8);
    printf("%s: %d CUs, clock %d MHz, %d one-wave blocks (16 per CU), %d wave instructions each\n", p.name, cus, p.clockRate / 1000, blocks, ITERS);
    run<RD32, false>("read b32", out, cyc, blocks, cus);   run<RD32, true>("read b32", out, cyc, blocks, cus);
    run<RD64, false>("read b64", out, cyc, blocks, cus);   run<RD64, true>("read b64", out, cyc, blocks, cus);
    run<WR32, false>("write b32", out, cyc, blocks, cus);  run<WR32, true>("write b32", out, cyc, blocks, cus);
    run<WR16, false>("write b16", out, cyc, blocks, cus);  run<WR16, true>("write b16", out, cyc, blocks, cus);
    run<ADD, false>("add", out, cyc, blocks, cus);         run<ADD, true>("add", out, cyc, blocks, cus);
    run<ADD_RTN, false>("add rtn", out, cyc, blocks, cus); run<ADD_RTN, true>("add rtn", out, cyc, blocks, cus);
    run<MIN_RTN, false>("min rtn", out, cyc, blocks, cus); run<MIN_RTN, true>("min rtn", out, cyc, blocks, cus);
    run<CAS_RTN, false>("cas rtn", out, cyc, blocks, cus); run<CAS_RTN, true>("cas rtn", out, cyc, blocks, cus);
    return 0;
}
