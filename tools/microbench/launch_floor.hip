// What does ONE launch of 2 040 one-wave workgroups (a 1080p picture: BASELINE config 3 as written) cost before any arithmetic?
// Times, by HIP events over a train of back-to-back launches on one stream (as bench.py's single-stream leg does):
//   empty        grid x 64 threads, the kernel returns at once                      -> launch + dispatch floor
//   header       every wave reads its 128-byte chunk header by scalar loads (cold: another array every launch) and stores a sum
//   chain        ... then 54 lanes load 16 bytes each at an address that depends on the header (a prediction window), x4, and
//                the wave stores 1.5 KB: the dependent chain of recon_kernel (header -> windows -> store) without its arithmetic
// for grids of 2 040 (one chunk of 4 macroblocks per wave) and 8 160 (one macroblock per wave: every wave resident too, four
// times the workgroups to dispatch).   usage: launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(64) void k_empty(const uint32_t *, const uint8_t *, uint8_t *) {}

__global__ __launch_bounds__(64) void k_header(const uint32_t *chunks, const uint8_t *, uint8_t *out)
{
    const uint32_t *h = chunks + (size_t)blockIdx.x * 32;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 32; i++)
        s += __builtin_nontemporal_load(h + i) * 0 + h[i];
    if (threadIdx.x == 0)
        reinterpret_cast<uint32_t *>(out)[blockIdx.x] = s;
}

template <int kWindows> __global__ __launch_bounds__(64) void k_chain(const uint32_t *chunks, const uint8_t *frames, uint8_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4672];
    const uint32_t *h = reinterpret_cast<const uint32_t *>(__builtin_assume_aligned(chunks + (size_t)blockIdx.x * 32, 128));
    const uint32_t lane = threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < kWindows; m++) {
        const uint32_t off = __builtin_amdgcn_readfirstlane(h[8 + 6 * m + 1]); // the record's window offset (scalar load)
        if (lane < 54) {
            const uint4 v = *reinterpret_cast<const uint4 *>(frames + off + lane * 16);
            *reinterpret_cast<uint4 *>(lds + 192 + m * 864 + lane * 16) = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kWindows; m++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(lds + 192 + m * 864 + (lane & 31) * 16);
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    // 384 bytes per macroblock leave (lanes 0..23 x 16 bytes each)
    if (lane < 24 * kWindows)
        *reinterpret_cast<uint4 *>(out + ((size_t)blockIdx.x * 24 * kWindows + lane) * 16) = acc;
}

template <class K> static void run(const char *name, K kern, int grid, const std::vector<uint32_t *> &chunk_sets, const uint8_t *frames, uint8_t *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int n = 200;
    for (int i = 0; i < 40; i++) // clocks up
        kern<<<grid, 64>>>(chunk_sets[i % chunk_sets.size()], frames, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < n; i++)
        kern<<<grid, 64>>>(chunk_sets[i % chunk_sets.size()], frames, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s grid %5d: %6.2f us per launch\n", name, grid, ms * 1e3 / n);
}

int main()
{
    const size_t frame_bytes = 3164160ull * 3; // three 1080p slots
    uint8_t *frames, *out;
    (void)hipMalloc(&frames, frame_bytes + 65536);
    (void)hipMalloc(&out, 8160 * 384 * 4);
    (void)hipMemset(frames, 1, frame_bytes + 65536);
    // 64 chunk arrays (cycled: a launch does not find its headers in L2 from the launch before), window offsets as a 1080p P picture
    // has them: macroblock k's window near its own position in the reference slot
    std::vector<uint32_t *> sets;
    for (int s = 0; s < 64; s++) {
        std::vector<uint32_t> h(8160 * 32, 0);
        for (int c = 0; c < 8160; c++)
            for (int m = 0; m < 4; m++) {
                const uint32_t mb = (uint32_t)((c * 4 + m) % 8160);
                h[(size_t)c * 32 + 8 + 6 * m + 1] = 3164160u * (uint32_t)(s % 2) + mb * 256 + ((mb * 7 + s) % 16) * 16;
            }
        uint32_t *d;
        (void)hipMalloc(&d, h.size() * 4);
        (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        sets.push_back(d);
    }
    for (int grid : {2040, 8160}) {
        run("empty kernel", k_empty, grid, sets, frames, out);
        run("header: 32 scalar dwords per wave, cold", k_header, grid, sets, frames, out);
        if (grid == 2040)
            run("chain: header -> 4 windows -> 1.5 KB stored", k_chain<4>, grid, sets, frames, out);
        else
            run("chain: header -> 1 window -> 384 B stored", k_chain<1>, grid, sets, frames, out);
    }
    return 0;
}
