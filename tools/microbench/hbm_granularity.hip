// What does the SIZE of a contiguous piece cost in HBM bandwidth?  Every wave reads 1 KB per load instruction (16 bytes
// per lane), as 1024 / K pieces of K contiguous bytes at pseudo-random K-aligned places of a buffer far larger than
// the L2s (K > 1024: consecutive loads walk on through the piece).  The reconstruction kernel's prediction windows are such
// reads: today 6 places per window (luma tile pairs of 512 B, Cb and Cr block pairs of 128 B).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/hbm_granularity.hip -o /tmp/hbm_granularity && /tmp/hbm_granularity
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <initializer_list>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// kLoads independent loads in flight per wave and iteration
template <int kLoads>
__global__ __launch_bounds__(256) void read_pieces(const uint4 *buf, uint64_t n_bytes, uint32_t piece, uint32_t iters, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_pieces = n_bytes / piece;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[kLoads];
#pragma unroll
        for (int k = 0; k < kLoads; k++) {
            const uint64_t load = (wave * iters + it) * kLoads + k; // this wave's load number: 1 KB each
            uint64_t off;
            if (piece >= 1024) { // the piece is walked through by piece / 1024 consecutive loads
                const uint64_t per = piece / 1024;
                off = (mix(load / per) % n_pieces) * piece + (load % per) * 1024 + lane * 16;
            } else {             // 1024 / piece pieces per load
                const uint32_t lanes_per = piece / 16;
                off = (mix(load * 64 + lane / lanes_per) % n_pieces) * piece + (lane % lanes_per) * 16;
            }
            v[k] = buf[off / 16];
        }
#pragma unroll
        for (int k = 0; k < kLoads; k++)
            acc ^= v[k].x ^ v[k].w;
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

// the reconstruction kernel's mix: per iteration 4 KB read as 256-byte pieces at random places + 2 KB written as whole
// contiguous kilobytes (every wave walks its own output range)
__global__ __launch_bounds__(256) void read_pieces_write_rows(const uint4 *buf, uint64_t n_bytes, uint4 *out, uint32_t iters, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_pieces = n_bytes / 256;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t load = (wave * iters + it) * 4 + k;
            v[k] = buf[((mix(load * 64 + lane / 16) % n_pieces) * 256 + (lane % 16) * 16) / 16];
        }
        uint4 *o = out + ((wave * iters + it) * 2048) / 16 + lane;
        o[0] = v[0];
        o[64] = v[1];
        acc ^= v[2].x ^ v[3].w;
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

int main()
{
    const uint64_t n_bytes = 16ull << 30;
    uint4 *buf;
    uint32_t *sink;
    CK(hipMalloc(&buf, n_bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, n_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t blocks = 256 * 8 * 4, iters = 64; // 8 waves per SIMD resident, 4 rounds of workgroups
    for (uint32_t piece : {64u, 128u, 256u, 512u, 1024u, 2048u, 4096u, 16384u}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(read_pieces<4>, dim3(blocks), dim3(256), 0, 0, buf, n_bytes, piece, iters, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double bytes = (double)blocks * 4 * iters * 4 * 1024;
        printf("pieces of %5u contiguous bytes at random places: %7.1f GB/s (%.3f ms for %.2f GB)\n", piece, bytes / best / 1e6, best, bytes / 1e9);
    }
    {
        uint4 *out;
        const uint64_t out_bytes = (uint64_t)blocks * 4 * iters * 2048;
        CK(hipMalloc(&out, out_bytes));
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(read_pieces_write_rows, dim3(blocks), dim3(256), 0, 0, buf, n_bytes, out, iters, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double bytes = (double)blocks * 4 * iters * (4096 + 2048);
        printf("4 KB read as 256-byte pieces at random places + 2 KB written contiguously, per wave and iteration: %7.1f GB/s read + written (%.3f ms for %.2f GB)\n",
               bytes / best / 1e6, best, bytes / 1e9);
    }
    return 0;
}
