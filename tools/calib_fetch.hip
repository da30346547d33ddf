// tools/calib_fetch.hip -- calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths of the walk kernel
// (MI355X_MICROARCH.md: FETCH_SIZE is calibrated only for 16 B/lane streams, where it reads exactly 1/2).
//   k_stream8  : every lane reads 8 B, coalesced, over a 2 GiB buffer once            -> known bytes = 2 GiB
//   k_stream16 : every lane reads 16 B, coalesced, same buffer                         -> known bytes = 2 GiB (reference point)
//   k_rows8    : one wave per random row of 40 int2 (320 B, 8-byte aligned rows of a 1 GiB array), 4 Mi rows -> 1.25 GiB useful
// build + run:  hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib && rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- /tmp/calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_stream8(const int2 *p, size_t n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { int2 v = p[i]; acc += (unsigned)v.x + (unsigned)v.y; }
    if (acc == 0x123456789ull) out[0] = acc;
}
__global__ void k_stream16(const int4 *p, size_t n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { int4 v = p[i]; acc += (unsigned)v.x + (unsigned)v.y + (unsigned)v.z + (unsigned)v.w; }
    if (acc == 0x123456789ull) out[0] = acc;
}
__global__ void k_rows8(const int2 *p, size_t n_entries, size_t rows, unsigned long long *out) {
    const int lane = threadIdx.x & 63;
    unsigned long long acc = 0;
    for (size_t r = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * (blockDim.x >> 6)) {
        unsigned long long h = (r + 1) * 0x9e3779b97f4a7c15ull; h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
        const size_t start = (size_t)(h % (n_entries - 64));
        if (lane < 40) { int2 v = p[start + lane]; acc += (unsigned)v.x; }
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
// one lane per random 128-byte line of a 2 GiB array: A reads 8 B at the start of the line; B also reads 8 B from the line's second
// 64-byte half.  If single gathers were served by 64-byte requests B would count twice A; if a miss always brings the whole
// 128-byte line (tallied as 64 B) B counts the same as A.
__global__ void k_gather(const int2 *p, size_t n_lines, size_t gathers, int both, unsigned long long *out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < gathers; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = (i + 1) * 0x9e3779b97f4a7c15ull; h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
        const size_t line = (size_t)(h % n_lines);
        int2 v = p[line * 16]; acc += (unsigned)v.x;
        if (both) { int2 u = p[line * 16 + 8]; acc += (unsigned)u.y; }
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    void *buf; unsigned long long *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 8)); CK(hipMemset(buf, 1, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_stream8, dim3(4096), dim3(256), 0, 0, (const int2 *)buf, bytes / 8, out);
        hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, (const int4 *)buf, bytes / 16, out);
        hipLaunchKernelGGL(k_rows8, dim3(4096), dim3(256), 0, 0, (const int2 *)buf, ((size_t)1 << 30) / 8, (size_t)4 << 20, out);
        hipLaunchKernelGGL(k_gather, dim3(4096), dim3(256), 0, 0, (const int2 *)buf, bytes / 128, (size_t)8 << 20, 0, out);
        hipLaunchKernelGGL(k_gather, dim3(4096), dim3(256), 0, 0, (const int2 *)buf, bytes / 128, (size_t)8 << 20, 1, out);
    }
    CK(hipDeviceSynchronize());
    printf("k_gather: 8 Mi gathers, each from its own random 128-B line (first launch: 8 B per line, second: 8 B from each 64-B half)\n");
    printf("known bytes: k_stream8 %zu  k_stream16 %zu  k_rows8 useful %zu (4Mi rows x 320 B; 3-4 128-B lines per row = 1.5-2.0 GiB of lines)\n", bytes, bytes, ((size_t)4 << 20) * 320);
    return 0;
}
