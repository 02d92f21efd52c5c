// Is the data of global_load_lds_dwordx4 in LDS once s_waitcnt vmcnt(0) lets the wave go?  Many waves, cold lines.
// mode bit 0: wait with vmcnt(N = loads still allowed in flight) per load (in-order assumption), else vmcnt(0)
// mode bit 1: a regular VGPR load is issued FIRST and waited for with vmcnt(4) (mixed types)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ __launch_bounds__(256) void probe(const uint8_t *src, uint64_t span, uint32_t *bad, int mode)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[4 * 4096];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t *lds = lds_all + w * 4096;
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + w;
    // 4 loads, each lane its own 16 bytes, rows 7936 bytes apart (cold lines)
    const uint8_t *p0 = src + (wave * 1315423911ull) % (span - (1u << 20));
    {
        const uint64_t u = (uintptr_t)p0 & ~(uintptr_t)3;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        p0 = (const uint8_t *)(((uint64_t)hi << 32) | lo);
    }
    uint32_t first = 0;
    const uint32_t *fp = (const uint32_t *)(p0 + 512 * 1024) + lane;
    if (mode & 2)
        asm volatile("global_load_dword %0, %1, off" : "=v"(first) : "v"(fp) : "memory");
    for (int k = 0; k < 4; k++) {
        const uint32_t off = (uint32_t)lane * 7936 + k * 16;
        const uint32_t m0 = base + k * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(p0), "s"(m0) : "memory");
    }
    uint32_t wrong = 0;
    if (mode & 2) {
        __builtin_amdgcn_s_waitcnt(0x0f74);
        asm volatile("" : "+v"(first)::"memory");
        wrong += first != *fp ? 0x10000u : 0u;
    }
    for (int k = 0; k < 4; k++) {
        if (mode & 1) {
            if (k == 0) __builtin_amdgcn_s_waitcnt(0x0f73);
            if (k == 1) __builtin_amdgcn_s_waitcnt(0x0f72);
            if (k == 2) __builtin_amdgcn_s_waitcnt(0x0f71);
            if (k == 3) __builtin_amdgcn_s_waitcnt(0x0f70);
        } else if (k == 0) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
        }
        asm volatile("" ::: "memory");
        const volatile uint32_t *g = reinterpret_cast<volatile uint32_t *>(lds + k * 1024 + lane * 16);
        const uint32_t g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
        const uint32_t *q = (const uint32_t *)(p0 + (uint32_t)lane * 7936 + k * 16);
        wrong += (g0 != q[0]) + (g1 != q[1]) + (g2 != q[2]) + (g3 != q[3]);
    }
    if (wrong)
        atomicAdd(&bad[(lane >> 5) + 2 * ((wrong >> 16) ? 1 : 0)], wrong & 0xffff ? 1u : 0u), atomicAdd(&bad[4], wrong >> 16);
}
int main()
{
    const uint64_t span = 1ull << 30;
    uint8_t *d;
    uint32_t *bad;
    (void)hipMalloc(&d, span);
    (void)hipMalloc(&bad, 64);
    std::vector<uint32_t> h(span / 4);
    for (size_t i = 0; i < h.size(); i++)
        h[i] = (uint32_t)(i * 2654435761u) | 1u; // never 0
    (void)hipMemcpy(d, h.data(), span, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; mode++) {
        (void)hipMemset(bad, 0, 64);
        hipLaunchKernelGGL(probe, dim3(200000), dim3(256), 0, 0, d, span, bad, mode);
        (void)hipDeviceSynchronize();
        uint32_t r[8];
        (void)hipMemcpy(r, bad, 32, hipMemcpyDeviceToHost);
        printf("mode %d (per-load vmcnt(N) %d, VGPR load first %d): lanes with stale LDS data: first half %u, second half %u; stale VGPR-load lanes %u\n",
               mode, mode & 1, (mode >> 1) & 1, r[0] + r[2], r[1] + r[3], r[4]);
    }
    return 0;
}
