// sync_probe.hip -- diagnostic only: what a host round trip costs on this stack, for the latency budget of the device batch pass
// (DESIGN section 9 N4): kernel launch + hipStreamSynchronize against launch + polling a flag the kernel writes to pinned host
// memory; a 20 KB pinned H2D copy in front; a kernel that READS 20 KB of pinned host memory instead of the copy.
// build: hipcc --offload-arch=gfx950 -O2 tools/sync_probe.hip -o ab/sync_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>
__global__ void k_flag(volatile uint32_t *flag, uint32_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) { *flag = v; } }
__global__ void k_sum(const uint64_t *src, int n, volatile uint32_t *flag, uint32_t v, uint64_t *sink) {
    uint64_t s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += src[i];
    atomicAdd((unsigned long long *)sink, (unsigned long long)s);
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); *flag = v; }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static void report(const char *name, F f) {
    std::vector<double> t;
    for (int i = 0; i < 300; ++i) { const double a = now(); f(i); t.push_back(now() - a); }
    std::sort(t.begin(), t.end());
    std::printf("%-62s median %7.2f us   p10 %7.2f   p90 %7.2f\n", name, t[150], t[30], t[270]);
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t *flag; hipHostMalloc((void **)&flag, 64, hipHostMallocDefault); *flag = 0;
    const int n = 2560;                                   // 20 KB
    uint64_t *hsrc, *dsrc, *sink; hipHostMalloc((void **)&hsrc, n * 8, hipHostMallocDefault); hipMalloc((void **)&dsrc, n * 8); hipMalloc((void **)&sink, 8);
    for (int i = 0; i < n; ++i) hsrc[i] = i;
    uint32_t tick = 1;
    report("empty kernel + hipStreamSynchronize", [&](int) { hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, flag, tick++); hipStreamSynchronize(s); });
    report("empty kernel + poll the pinned flag", [&](int) { const uint32_t v = tick++; hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, flag, v); while (*(volatile uint32_t *)flag != v) {} });
    report("20 KB H2D (pinned) + kernel over it + hipStreamSynchronize", [&](int) { hipMemcpyAsync(dsrc, hsrc, n * 8, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, s, dsrc, n, flag, tick++, sink); hipStreamSynchronize(s); });
    report("20 KB H2D (pinned) + kernel over it + poll", [&](int) { const uint32_t v = tick++; hipMemcpyAsync(dsrc, hsrc, n * 8, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, s, dsrc, n, flag, v, sink); while (*(volatile uint32_t *)flag != v) {} });
    report("kernel reading 20 KB of pinned host memory + poll", [&](int) { const uint32_t v = tick++; hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, s, hsrc, n, flag, v, sink); while (*(volatile uint32_t *)flag != v) {} });
    report("8-byte D2H (pinned) + hipStreamSynchronize", [&](int) { hipMemcpyAsync(hsrc, sink, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); });
    hipDeviceSynchronize();
    return 0;
}
