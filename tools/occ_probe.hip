// tools/occ_probe.hip -- how many one-wave workgroups per CU fit for a given LDS size (allocation granularity probe)
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned int dyn[];
__global__ __launch_bounds__(64, 8) void k(unsigned int *out) { dyn[threadIdx.x] = threadIdx.x; __syncthreads(); out[threadIdx.x] = dyn[63 - threadIdx.x]; }
__global__ __launch_bounds__(128, 8) void k2(unsigned int *out) { dyn[threadIdx.x] = threadIdx.x; __syncthreads(); out[threadIdx.x] = dyn[127 - threadIdx.x]; }
int main() {
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int bytes : {6144, 6656, 6826, 7168, 7600, 7680, 7936, 8112, 8192, 8704, 8960, 9216, 10240, 10752, 16224, 16384}) {
        int n1 = 0, n2 = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, k, 64, bytes);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, k2, 128, bytes);
        printf("LDS %6d B/block: %2d blocks of 64 threads, %2d blocks of 128 threads per CU\n", bytes, n1, n2);
    }
    return 0;
}
