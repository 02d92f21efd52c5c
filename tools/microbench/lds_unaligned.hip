// Does gfx950 read LDS at any BYTE alignment in hardware, and what does it cost?
// The compiler says yes (a packed 4-byte LDS load compiles to ONE ds_read_b32: amdhsa runs in unaligned access mode);
// this probe checks the bytes that come back and times three ways of fetching 4 pixels that start `x` bytes into a
// 32-byte window row, in the reconstruction kernel's lane map (lane = row lane>>2, quarter lane&3, one wave = 16 rows):
//   A  ds_read2_b32 at the dword below + v_alignbit_b32 by 8 (x & 3)        (what recon_kernel's rc_mc_luma does today)
//   B  ONE ds_read_b32 at the byte address                                 (unaligned)
//   C  the 4-tap form: A = 2 x ds_read2_b32 + 4 funnel shifts; B = 4 x ds_read_b32 at byte addresses
// usage: lds_unaligned  (prints correctness for every x in 0..15 and ns per wave-iteration)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__device__ __forceinline__ uint32_t ds_u32(uint32_t addr)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

__global__ __launch_bounds__(64) void check(uint32_t *out, int x)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64)
        lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    const uint32_t base = (uint32_t)(uintptr_t)lds;
    out[lane] = ds_u32(base + (lane >> 2) * 32 + (lane & 3) * 4 + (uint32_t)x);
}

// the timed kernels, written with plain asm blocks (no 64-bit register tricks)
__global__ __launch_bounds__(64) void copy_aligned(uint32_t *out, uint32_t x, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4672];
    for (int i = threadIdx.x; i < 4672; i += 64)
        lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    const uint32_t base = (uint32_t)(uintptr_t)lds + (lane >> 2) * 32 + (lane & 3) * 4;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint32_t xx = __builtin_amdgcn_readfirstlane((x + (uint32_t)it) & 15u);
        const uint32_t addr = base + (xx & 12u);
        uint32_t p0, p1;
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p0), "=&v"(p1) : "v"(addr) : "memory");
        acc += __builtin_amdgcn_alignbit(p1, p0, (xx & 3u) * 8u);
    }
    out[blockIdx.x * 64 + lane] = acc;
}
__global__ __launch_bounds__(64) void copy_aligned2(uint32_t *out, uint32_t x, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4672];
    for (int i = threadIdx.x; i < 4672; i += 64)
        lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    const uint32_t base = (uint32_t)(uintptr_t)lds + (lane >> 2) * 32 + (lane & 3) * 4;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint32_t xx = __builtin_amdgcn_readfirstlane((x + (uint32_t)it) & 15u);
        const uint32_t addr = base + (xx & 12u);
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        u2 p;
        asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p) : "v"(addr) : "memory");
        acc += __builtin_amdgcn_alignbit(p.y, p.x, (xx & 3u) * 8u);
    }
    out[blockIdx.x * 64 + lane] = acc;
}
__global__ __launch_bounds__(64) void copy_unaligned(uint32_t *out, uint32_t x, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4672];
    for (int i = threadIdx.x; i < 4672; i += 64)
        lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    const uint32_t base = (uint32_t)(uintptr_t)lds + (lane >> 2) * 32 + (lane & 3) * 4;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint32_t xx = __builtin_amdgcn_readfirstlane((x + (uint32_t)it) & 15u);
        const uint32_t addr = base + xx;
        uint32_t p0;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p0) : "v"(addr) : "memory");
        acc += p0;
    }
    out[blockIdx.x * 64 + lane] = acc;
}
__global__ __launch_bounds__(64) void bilin_aligned(uint32_t *out, uint32_t x, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4672];
    for (int i = threadIdx.x; i < 4672; i += 64)
        lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    const uint32_t base = (uint32_t)(uintptr_t)lds + (lane >> 2) * 32 + (lane & 3) * 4;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint32_t xx = __builtin_amdgcn_readfirstlane((x + (uint32_t)it) & 15u);
        const uint32_t addr = base + (xx & 12u);
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        u2 p, q;
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:8 offset1:9\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p), "=&v"(q) : "v"(addr) : "memory");
        const uint32_t s = (xx & 3u) * 8u;
        const uint64_t a = (uint64_t)p.x | ((uint64_t)p.y << 32), b = (uint64_t)q.x | ((uint64_t)q.y << 32);
        acc += (uint32_t)(a >> s) ^ (uint32_t)(a >> (s + 8)) ^ (uint32_t)(b >> s) ^ (uint32_t)(b >> (s + 8));
    }
    out[blockIdx.x * 64 + lane] = acc;
}
__global__ __launch_bounds__(64) void bilin_unaligned(uint32_t *out, uint32_t x, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4672];
    for (int i = threadIdx.x; i < 4672; i += 64)
        lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    const uint32_t base = (uint32_t)(uintptr_t)lds + (lane >> 2) * 32 + (lane & 3) * 4;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint32_t xx = __builtin_amdgcn_readfirstlane((x + (uint32_t)it) & 15u);
        const uint32_t addr = base + xx;
        uint32_t a, b, c, d;
        asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:1\n\tds_read_b32 %2, %4 offset:32\n\tds_read_b32 %3, %4 offset:33\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                     : "v"(addr)
                     : "memory");
        acc += a ^ b ^ c ^ d;
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <class K> static float time_it(K kern, uint32_t *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<blocks, 64>>>(d, 1, 16); // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 64>>>(d, 1, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    uint32_t *d;
    hipMalloc(&d, 64 * 8192 * 4);
    uint32_t h[64];
    int bad = 0;
    for (int x = 0; x < 16; x++) {
        check<<<1, 64>>>(d, x);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int lane = 0; lane < 64; lane++) {
            const uint32_t at = (lane >> 2) * 32 + (lane & 3) * 4 + x;
            uint32_t want = 0;
            for (int k = 0; k < 4; k++)
                want |= (uint32_t)(uint8_t)((at + k) * 7 + 3) << (8 * k);
            if (h[lane] != want) {
                if (bad < 8)
                    printf("x = %d lane %d: got %08x want %08x\n", x, lane, h[lane], want);
                bad++;
            }
        }
    }
    printf("unaligned ds_read_b32, the 4 bytes at the byte address: %s (%d mismatches over x = 0..15 x 64 lanes)\n", bad ? "WRONG" : "exact", bad);
    // throughput: 8192 one-wave workgroups (8 waves per SIMD resident), 4096 iterations each
    const int blocks = 8192, iters = 4096;
    const double waves_iters = (double)blocks * iters;
    struct { const char *name; float ms; } r[] = {
        {"copy: 2 x ds_read_b32 aligned + v_alignbit", time_it(copy_aligned, d, blocks, iters)},
        {"copy: ds_read2_b32 aligned + v_alignbit   ", time_it(copy_aligned2, d, blocks, iters)},
        {"copy: 1 x ds_read_b32 at the byte address ", time_it(copy_unaligned, d, blocks, iters)},
        {"4 taps: 2 x ds_read2_b32 + 4 64-bit shifts ", time_it(bilin_aligned, d, blocks, iters)},
        {"4 taps: 4 x ds_read_b32 at byte addresses  ", time_it(bilin_unaligned, d, blocks, iters)},
    };
    for (auto &k : r)
        printf("%s  %8.3f ms  = %.2f ns per wave-iteration per CU-slot (%.1f clocks at 2.4 GHz per wave-iteration, 32 waves per CU)\n", k.name, k.ms,
               k.ms * 1e6 / waves_iters * 256, k.ms * 1e6 / waves_iters * 256 * 2.4);
    return bad != 0;
}
