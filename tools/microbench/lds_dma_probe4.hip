// global_load_lds_dwordx4 with an instruction offset: which global bytes land where in LDS?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void probe(const uint8_t *src, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    for (int i = threadIdx.x; i < 1024; i += 64)
        reinterpret_cast<uint32_t *>(lds)[i] = 0xEEEEEEEEu;
    __syncthreads();
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    const uint32_t off = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1312" ::"v"(off), "s"(src), "s"(base) : "memory");
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64)
        out[i] = reinterpret_cast<uint32_t *>(lds)[i];
}
int main()
{
    std::vector<uint32_t> h(2048);
    for (int i = 0; i < 2048; i++)
        h[i] = i;
    uint8_t *d;
    uint32_t *o;
    (void)hipMalloc(&d, 8192);
    (void)hipMalloc(&o, 4096);
    (void)hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    std::vector<uint32_t> r(1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int first = -1;
    for (int i = 0; i < 1024; i++)
        if (r[i] != 0xEEEEEEEEu) { first = i; break; }
    printf("offset:1312 (= 328 dwords): first written LDS dword %d holds source dword %u; next %u %u %u; lds dword %d = %x\n", first, first >= 0 ? r[first] : 0,
           first >= 0 ? r[first + 1] : 0, first >= 0 ? r[first + 2] : 0, first >= 0 ? r[first + 3] : 0, first + 255, first >= 0 ? r[first + 255] : 0);
    return 0;
}
