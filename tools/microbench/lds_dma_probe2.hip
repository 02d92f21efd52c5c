// global_load_lds_dwordx4 in the saddr form (scalar base + 32-bit lane offset), with an exec mask, several in flight
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void probe(const uint8_t *src, uint32_t *out, int mode)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    for (int i = threadIdx.x; i < 1024; i += 64)
        reinterpret_cast<uint32_t *>(lds)[i] = 0xEEEEEEEEu;
    __syncthreads();
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    const int lane = threadIdx.x;
    for (int k = 0; k < 4; k++) {
        const uint32_t off = (uint32_t)lane * 16 + k * 1024 + (mode & 2 ? 4 : 0);
        const uint32_t m0 = base + k * 832;
        if (!(mode & 1) || lane < 52)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(src), "s"(m0) : "memory");
    }
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
    for (int mode = 0; mode < 4; mode++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, mode);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < ((mode & 1) ? 52 : 64); l++)
                for (int j = 0; j < 4; j++) {
                    const uint32_t want = (l * 16 + k * 1024 + ((mode & 2) ? 4 : 0)) / 4 + j;
                    const uint32_t at = (k * 832 + l * 16) / 4 + j;
                    // later loads overwrite the tail of earlier ones when all 64 lanes are on: check only the last writer
                    if ((mode & 1) || k == 3 || l < 52)
                        bad += r[at] != want;
                }
        printf("mode %d (mask %d, dword-misaligned source %d): %d wrong dwords; lds[208..212] = %x %x %x %x, lds[0..3] = %x %x %x %x\n", mode, mode & 1,
               (mode >> 1) & 1, bad, r[208], r[209], r[210], r[211], r[0], r[1], r[2], r[3]);
    }
    return 0;
}
