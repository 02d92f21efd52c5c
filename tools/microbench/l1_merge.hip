// Does the L1 (TCP) merge same-line 16-byte loads of NON-adjacent lanes?  Two lane mappings over the
// same addresses: mode 0 = lane (half, row) = (lane>>4, lane&15)  [the two halves of a row 16 lanes apart]
//                 mode 1 = lane = 2*row + half                     [the two halves in adjacent lanes]
//                 mode 2 = mode 0 with 8 lanes apart (the kernel's block layout: lane = b*8+j)
// Run under rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum and compare per-dispatch values.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
struct u4 { uint32_t v[4]; };
template <int MODE> __global__ void k(const uint8_t *buf, uint32_t stride, uint32_t *sink, uint32_t misalign)
{
    const int lane = threadIdx.x & 63;
    int row, half;
    if (MODE == 0) { half = (lane >> 4) & 1; row = lane & 15; }
    else if (MODE == 1) { row = (lane >> 1) & 15; half = lane & 1; }
    else { const int b = lane >> 3, j = lane & 7; half = b & 1; row = ((b >> 1) & 1) * 8 + j; }
    if (lane >= 32) return;
    const uint64_t base = ((uint64_t)blockIdx.x * 2654435761u % (1u << 20)) * 64 + misalign;
    const uint8_t *p = buf + base + (uint64_t)row * stride + half * 8;
    u4 x;
    __builtin_memcpy(&x, p, 16);
    if ((x.v[0] ^ x.v[1] ^ x.v[2] ^ x.v[3]) == 0x12345678u) sink[lane] = 1;
}
int main(int argc, char **argv)
{
    const uint32_t misalign = argc > 1 ? atoi(argv[1]) : 5;
    uint8_t *buf; uint32_t *sink;
    const size_t bytes = (size_t)(1u << 20) * 64 + 32 * 2048 + 4096;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&sink, 256);
    const int blocks = 1 << 18;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, buf, 1920u, sink, misalign);
        hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, buf, 1920u, sink, misalign);
        hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, buf, 1920u, sink, misalign);
    }
    hipDeviceSynchronize();
    printf("done %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
