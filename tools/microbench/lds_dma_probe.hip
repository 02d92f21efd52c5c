// Where do the 12 bytes per lane of global_load_lds_dwordx3 land in LDS?  (and dwordx4 for comparison)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
template <int SIZE> __global__ void probe(const uint8_t *src, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[2048];
    for (int i = threadIdx.x; i < 512; i += 64)
        reinterpret_cast<uint32_t *>(lds)[i] = 0xEEEEEEEEu;
    __syncthreads();
    if (SIZE == 12)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + threadIdx.x * 12),
                                         (__attribute__((address_space(3))) void *)lds, 12, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + threadIdx.x * 16),
                                         (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64)
        out[i] = reinterpret_cast<uint32_t *>(lds)[i];
}
int main()
{
    std::vector<uint32_t> h(1024);
    for (int i = 0; i < 1024; i++)
        h[i] = i; // dword i holds i
    uint8_t *d;
    uint32_t *o;
    hipMalloc(&d, 4096);
    hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    std::vector<uint32_t> r(512);
    for (int size : {12, 16}) {
        if (size == 12)
            hipLaunchKernelGGL(probe<12>, dim3(1), dim3(64), 0, 0, d, o);
        else
            hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), 0, 0, d, o);
        hipDeviceSynchronize();
        hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
        printf("size %d: LDS dwords 0..47:", size);
        for (int i = 0; i < 48; i++)
            printf(" %x", r[i]);
        printf("\n   dwords 180..200:");
        for (int i = 180; i < 200; i++)
            printf(" %x", r[i]);
        printf("\n   dwords 250..260:");
        for (int i = 250; i < 260; i++)
            printf(" %x", r[i]);
        printf("\n");
    }
    return 0;
}
