// How many 256-thread workgroups fit a CU for a given LDS size (allocation granularity of gfx950)?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(float *o) { extern __shared__ float s[]; s[threadIdx.x] = 1; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
int main()
{
    for (int b : {5120, 5664, 5792, 6144, 20480, 23168, 26000, 27000, 27306, 28000, 31000, 31744, 32000, 32256, 32512, 32632, 32768, 33000}) {
        int n = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, b);
        printf("LDS %6d B per workgroup of 256 threads: %d workgroups per CU\n", b, n);
    }
    return 0;
}
