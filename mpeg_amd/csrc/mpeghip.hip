// mpeghip.hip — gfx950 kernels and the C ABI of include/mpeghip.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
//        -I include -I mpeg_amd/csrc mpeg_amd/csrc/mpeghip.hip -o mpeg_amd/libmpeghip.so
//
// There is no CPU path in this library.  Every entry point that needs the GPU
// fails with MPEGHIP_ERR_NO_DEVICE / MPEGHIP_ERR_HIP when it is not there.
#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <memory>
#include <new>
#include <mutex>
#include <string>
#include <vector>

#include "audio_lane.h"
#include "iso11172_synth_window.h"
#include "mpeghip.h"
#include "video_lane.h"
#include "video_split_lane.h"
#include "video_compact_lane.h"
#include "video_wire_lane.h"

using namespace mpg;

// ============================================================ device kernels

// LDS ordering inside ONE wavefront: DS operations of a wave execute in issue
// order, so a ds_read that follows a ds_write sees it without an s_barrier; the
// fences only stop the compiler from moving accesses across the hand-off.
static __device__ __forceinline__ void wave_lds_handoff()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kWaveLdsBytes = kTileDwords * 4 + kRgbaBytes; // 1728 + 384

// One wavefront per macroblock at a time, WAVES macroblocks (consecutive
// descriptors, i.e. normally consecutive macroblocks of one row) per workgroup
// so that the 8-byte row stores of neighbouring macroblocks combine into full
// lines in one L2.
//
// MODE 0: one chunk (WAVES macroblocks) per workgroup, grid = all chunks.
// MODE 1: persistent workgroups; every XCD walks one contiguous range of chunks.
// MODE 2: MODE 1 + software pipeline: while macroblock i is computed, the global
//         loads of macroblock i+1 are in flight and the (scalar) descriptor of
//         macroblock i+2 is being fetched.  The kernel is latency-bound without
//         it: a wave's life is a chain of dependent round trips (descriptor ->
//         picture -> pixels/coefficients -> store).
template <int MODE>
static __device__ __forceinline__ void chunk_range(uint32_t n_chunks, uint32_t &first, uint32_t &last, uint32_t &step)
{
    if (MODE == 0) {
        first = xcd_chunk(blockIdx.x, gridDim.x);
        last = first + 1;
        step = 1;
    } else {
        const uint32_t nx = 8; // gridDim.x is a multiple of 8
        const uint32_t xcd = blockIdx.x % nx, k = blockIdx.x / nx, K = gridDim.x / nx;
        const uint32_t lo = (uint32_t)(((uint64_t)n_chunks * xcd) / nx);
        const uint32_t hi = (uint32_t)(((uint64_t)n_chunks * (xcd + 1)) / nx);
        first = lo + k;
        last = hi;
        step = K;
    }
}

static __device__ __forceinline__ void finish_mb(const VideoArgs &a, const MbU &u, int lane, const MbLane &st,
                                                 int32_t *tile, uint8_t *stage)
{
    wave_lds_handoff();
    bool wrote;
    const uint64_t out = mb_phase_b(a, u, lane, st, tile, wrote);
    if (u.rgba) { // wave-uniform
        mb_phase_c_stage(a, u, lane, out, wrote, stage);
        wave_lds_handoff();
        mb_phase_c_convert(a, u, lane, stage);
    }
    wave_lds_handoff(); // the tile is rewritten by this wave's next macroblock
}

template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void recon_kernel(const VideoArgs a, const uint32_t n_chunks)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[WAVES * kWaveLdsBytes];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int32_t *tile = reinterpret_cast<int32_t *>(lds + wave * kWaveLdsBytes);
    uint8_t *stage = lds + wave * kWaveLdsBytes + kTileDwords * 4;
    uint32_t first, last, step;
    chunk_range<MODE>(n_chunks, first, last, step);

    if (MODE == 3) {
        // Software pipeline with compile-time load / store counts (see mb_issue_loads_static):
        // two macroblocks' worth of loads are in flight per wave, the descriptor of a third
        // is being fetched by scalar loads.  Unrolled by two so the load registers ping-pong
        // without moves (a move would have to wait for the data).
        const uint32_t S = step * WAVES;
        uint32_t i = __builtin_amdgcn_readfirstlane(first * WAVES + wave);
        const uint64_t chunk_limit = (uint64_t)last * WAVES;
        const uint32_t limit = (uint32_t)(chunk_limit < a.n_mbs ? chunk_limit : a.n_mbs);
        if (i >= limit)
            return;
        uint8_t *sink = a.dump + ((uint64_t)blockIdx.x * WAVES + wave) * 512 + (uint32_t)lane * 8;
        auto clampi = [&](uint64_t x) { return __builtin_amdgcn_readfirstlane((uint32_t)(x < limit ? x : limit - 1)); };
        auto finish = [&](const MbU &u, const MbLane &st) {
            wave_lds_handoff();
            bool wrote;
            const uint64_t out = mb_phase_b_t<true>(a, u, lane, st, tile, wrote, sink);
            if (u.rgba) {
                mb_phase_c_stage(a, u, lane, out, wrote, stage);
                wave_lds_handoff();
                mb_phase_c_convert(a, u, lane, stage);
            }
            wave_lds_handoff();
        };
        MbU ua = load_mb(a, i);
        MbLoads la, lb;
        mb_issue_loads_static(a, ua, lane, la);
        MbU ub = load_mb(a, clampi((uint64_t)i + S));
        for (;;) {
            // even step: compute A while B's loads fly
            MbU uc = load_mb(a, clampi((uint64_t)i + 2ull * S));
            mb_issue_loads_static(a, ub, lane, lb);
            {
                MbLane st;
                mb_phase_a_compute_static(a, ua, lane, la, st, tile);
                finish(ua, st);
            }
            if ((uint64_t)i + S >= limit)
                break;
            // odd step: compute B while C's loads fly (C's loads land in A's registers)
            MbU ud = load_mb(a, clampi((uint64_t)i + 3ull * S));
            mb_issue_loads_static(a, uc, lane, la);
            {
                MbLane st;
                mb_phase_a_compute_static(a, ub, lane, lb, st, tile);
                finish(ub, st);
            }
            if ((uint64_t)i + 2ull * S >= limit)
                break;
            ua = uc;
            ub = ud;
            i += 2 * S;
        }
    } else if (MODE < 2) {
        for (uint32_t c = first; c < last; c += step) {
            const uint32_t mb_index = c * WAVES + wave;
            if (mb_index >= a.n_mbs)
                break;
            const MbU u = load_mb(a, mb_index);
            MbLane st;
            mb_phase_a(a, u, lane, st, tile);
            finish_mb(a, u, lane, st, tile, stage);
        }
    } else {
        // indices of this wave's macroblocks: i0, i0+S, i0+2S, ... while < limit (all wave-uniform)
        const uint32_t S = step * WAVES;
        uint32_t i = __builtin_amdgcn_readfirstlane(first * WAVES + wave);
        const uint64_t chunk_limit = (uint64_t)last * WAVES;
        const uint32_t limit = (uint32_t)(chunk_limit < a.n_mbs ? chunk_limit : a.n_mbs);
        if (i >= limit)
            return;
        MbU u0 = load_mb(a, i);
        MbLoads l0;
        mb_issue_loads(a, u0, lane, l0);
        // look-ahead indices are clamped instead of branched on: the descriptor loads stay scalar and unconditional
        MbU u1 = load_mb(a, __builtin_amdgcn_readfirstlane(i + S < limit ? i + S : i));
        for (;;) {
            const bool has1 = i + S < limit;
            const uint32_t i2 = (uint64_t)i + 2ull * S < limit ? i + 2 * S : i;
            const MbU u2 = load_mb(a, __builtin_amdgcn_readfirstlane(i2)); // scalar loads, two macroblocks ahead
            MbLoads l1;
            mb_issue_loads(a, u1, lane, l1); // vector loads of the next macroblock: in flight during the compute below
            MbLane st;
            mb_phase_a_compute(a, u0, lane, l0, st, tile);
            finish_mb(a, u0, lane, st, tile, stage);
            if (!has1)
                break;
            u0 = u1;
            l0 = l1;
            u1 = u2;
            i += S;
        }
    }
}

// ---- compact path (video_compact_lane.h): fused, single pass over the pixels, dense residual stage
__global__ __launch_bounds__(kChunkMbs * 64) void recon_compact_kernel(const VideoArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kCompactLdsBytes];
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint32_t chunk = xcd_chunk(blockIdx.x, gridDim.x);
    const ChunkInfo ci = load_chunk(a, chunk);   // scalar loads of the chunk's 8 descriptors
    const bool have_mb = w < ci.n;
    MbU u;
    MbLoads ld;
    if (have_mb) {
        u = load_mb(a, chunk * kChunkMbs + w);
        compact_phase1(a, u, lane, ld);          // prediction loads: in flight during phase 2
    }
    if (8 * w < ci.base[kChunkMbs]) {            // wave-uniform: this wave has coded blocks to transform
        const int g = lane >> 3, j = lane & 7;
        const uint32_t slot = 8 * w + (uint32_t)g;
        int32_t *tile_g = reinterpret_cast<int32_t *>(lds + kResidStoreBytes) + (w * 8 + (uint32_t)g) * kTileStride;
        bool active;
        compact_phase2(a, ci, slot, j, tile_g, active);
        wave_lds_handoff();
        compact_phase2_rows(slot, j, tile_g, active, lds);
    }
    __syncthreads();
    if (have_mb)
        compact_phase3(a, u, ci, w, lane, ld, lds);
}

// ---- wave-chunk path: one wave = 4 consecutive macroblocks, dense residual stage, no barrier
// 8 waves per SIMD: 64 VGPRs, and 4 x 5120 bytes of LDS per workgroup let 8 workgroups share a CU
// kRgba: the instance for batches with MPEGHIP_PIC_RGBA pictures (Frame.RGBA() fused); the other one carries
// none of that code
template <int WAVES, bool kRgba>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void recon_wc_kernel(const VideoArgs a, const uint32_t n_chunks)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[WAVES * kWcLdsBytes];
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint32_t chunk = __builtin_amdgcn_readfirstlane(xcd_chunk(blockIdx.x, gridDim.x) * WAVES + w);
    if (chunk >= n_chunks)
        return;
    uint8_t *resid = lds_all + w * kWcLdsBytes;
    int32_t *tile = reinterpret_cast<int32_t *>(resid + kWcResidBytes);
#ifdef MPG_PHASE_TIMING // instrumented build for tools/phase_timing.py only: s_memtime at the phase boundaries
#define MPG_STAMP(k) ts[k] = __builtin_readcyclecounter()
    uint64_t ts[6];
#else
#define MPG_STAMP(k)
#endif
    MPG_STAMP(0);
    uint32_t n_live;
    WcRaw raw;
    wc_load_raw<kRgba>(a, chunk, n_live, raw); // one round of scalar loads for the whole chunk
    const WcInfo ci = wc_info_from_raw(n_live, raw);
    const int g = lane >> 3, j = lane & 7;
    MPG_STAMP(1);

    // prediction loads of every macroblock of the chunk, up front
    MbU u[kWcMbs];
    MbLoads ld[kWcMbs];
#pragma unroll
    for (int m = 0; m < kWcMbs; m++) {
        u[m] = wc_mb_from_raw<kRgba>(a, raw.d[m]);
        wc_issue_pred(a, u[m], lane, ld[m]);
    }
    MPG_STAMP(2);
    // dense residual stage: 8 coded blocks per pass
    const uint32_t total = ci.base[kWcMbs];
    for (uint32_t s0 = 0; s0 < total; s0 += 8) {
        const uint32_t slot = s0 + (uint32_t)g;
        bool active;
        compact_phase2(a, ci, slot, j, tile + g * kWcTileStride, active);
        wave_lds_handoff();
        compact_phase2_rows(slot, j, tile + g * kWcTileStride, active, resid);
        wave_lds_handoff();
    }
    MPG_STAMP(3);
    // per macroblock: prediction + residual, clamp; outputs leave as whole rows when the chunk is a horizontal run
    const bool coalesce = wc_can_coalesce(ci, u);
    bool rgba = false; // any macroblock of a picture that is colour-converted on the fly (wave-uniform)
#pragma unroll
    for (int m = 0; m < kWcMbs; m++)
        rgba = rgba || (kRgba && (uint32_t)m < ci.n && u[m].rgba != nullptr);
    uint8_t *out_tile = (coalesce || rgba) ? reinterpret_cast<uint8_t *>(tile) : nullptr;
    const int below_lane = wc_below_lane(lane);
    const int below_addr = (below_lane < 0 ? lane : below_lane) << 2; // ds_bpermute byte address of the source lane
    auto row_below = [&](int m) { // the row under this lane's row: from the lane that loaded it, or ld.r1
        u8x16 below = ld[m].r1;
        if (wc_needs_below(u[m])) { // wave-uniform
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute(below_addr, (int)ld[m].r0.v[k]);
                below.v[k] = below_lane < 0 ? below.v[k] : got;
            }
        }
        return below;
    };
    if (coalesce) { // the normal case, without the rare paths
#pragma unroll
        for (int m = 0; m < kWcMbs; m++)
            wc_phase3<kWcMbs, true>(a, u[m], ci, (uint32_t)m, lane, ld[m], row_below(m), resid, out_tile, false);
    } else {
#pragma unroll
        for (int m = 0; m < kWcMbs; m++) {
            if ((uint32_t)m >= ci.n)
                continue;
            wc_phase3<kWcMbs, false>(a, u[m], ci, (uint32_t)m, lane, ld[m], row_below(m), resid, out_tile, true);
        }
    }
    MPG_STAMP(4);
    if (out_tile) {
        wave_lds_handoff();
        if (coalesce)
            wc_store_tile(a, u[0], lane, out_tile);
        if (kRgba && rgba) {
#pragma unroll
            for (int m = 0; m < kWcMbs; m++)
                if ((uint32_t)m < ci.n && u[m].rgba != nullptr)
                    wc_rgba_mb(a, u[m], (uint32_t)m, lane, out_tile);
        }
    }
#ifdef MPG_PHASE_TIMING
    MPG_STAMP(5);
    if (lane == 0 && chunk < 60000) { // a.dump is 4 MB: 64 bytes per sampled wave
        uint64_t *d = reinterpret_cast<uint64_t *>(a.dump) + (uint64_t)chunk * 8;
        for (int k = 0; k < 6; k++)
            d[k] = ts[k];
        d[6] = total;
        d[7] = coalesce;
    }
#endif
#undef MPG_STAMP
}

// ---- split path (video_split_lane.h): K1 prediction, K2 dense residual
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pred_kernel(const SplitArgs s)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint32_t chunk = xcd_chunk(blockIdx.x, gridDim.x);
    const uint32_t first = (chunk * WAVES + wave) * 2; // wave-uniform
    if (first >= s.v.n_mbs)
        return;
    const uint32_t second = first + 1 < s.v.n_mbs ? first + 1 : first;
    const PredMb m0 = load_pred_mb(s.v, first);   // scalar loads
    const PredMb m1 = load_pred_mb(s.v, second);
    const bool hi = lane >= 32;
    if (hi && first + 1 >= s.v.n_mbs)
        return;
    pred_lane(s, select_pred_mb(hi, m0, m1), lane & 31);
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void resid_kernel(const SplitArgs s)
{
    __shared__ __attribute__((aligned(16))) int32_t tiles[WAVES * kResidTileDwords];
    const uint32_t wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 3, j = lane & 7;
    const uint32_t chunk = xcd_chunk(blockIdx.x, gridDim.x);
    const uint32_t unit = (chunk * WAVES + wave) * 8 + (uint32_t)g;
    int32_t *tile_g = tiles + wave * kResidTileDwords + g * kTileStride;
    ResidLane st;
    resid_phase_a(s, unit, j, tile_g, st);
    wave_lds_handoff();
    resid_phase_b(s, j, tile_g, st);
}

// Frame.RGBA of the cur slot of every picture flagged MPEGHIP_PIC_RGBA: grid (x quads, rows, pictures).
__global__ __launch_bounds__(256) void rgba_pics_kernel(const VideoArgs a, uint32_t pic0)
{
    const mpeghip_pic_desc p = a.pics[pic0 + blockIdx.z];
    if (!(p.flags & MPEGHIP_PIC_RGBA))
        return;
    const uint32_t x4 = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const uint64_t fs = (uint64_t)p.stream * MPEGHIP_SLOTS + p.cur;
    rgba_convert_quad(a.frames + fs * a.frame_stride, a.luma_w, a.chroma_w, a.luma_bytes, a.chroma_bytes, a.width,
                      a.height, x4, y, a.rgba + fs * a.rgba_stride);
}

// Frame.RGBA for whole slots: grid (x quads, row pairs / 4, streams).
__global__ __launch_bounds__(256) void rgba_kernel(const uint8_t *frames, uint64_t frame_stride,
                                                  uint8_t *rgba, uint64_t rgba_stride,
                                                  uint32_t luma_w, uint32_t chroma_w,
                                                  uint32_t luma_bytes, uint32_t chroma_bytes,
                                                  uint32_t width, uint32_t height,
                                                  uint32_t slot, uint32_t stream0)
{
    const uint32_t x4 = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const uint64_t fs = (uint64_t)(stream0 + blockIdx.z) * MPEGHIP_SLOTS + slot;
    rgba_convert_quad(frames + fs * frame_stride, luma_w, chroma_w, luma_bytes, chroma_bytes,
                      width, height, x4, y, rgba + fs * rgba_stride);
}

// Replicate a one-stream descriptor set for streams 1..n-1 (benchmark batches): descriptors and pictures
// for the diagnostic kernels, expanded records (video_compact_lane.h) for the wave-chunk kernel.  Upload-time
// scaffolding of mpeghip_video_batch_upload_replicated, never inside a timed region.
struct ReplicateSteps {
    uint32_t coef_units;  // coefficient units of one stream
    uint32_t frames256;   // MPEGHIP_SLOTS * frame_stride >> 8
    uint32_t rgba256;     // MPEGHIP_SLOTS * rgba_stride >> 8
};
__global__ void replicate_desc_kernel(mpeghip_pic_desc *pics, uint32_t n_pics, mpeghip_mb_desc *mbs, uint32_t *xmbs,
                                      uint32_t n_mbs, ReplicateSteps k, uint32_t n_streams)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total_mbs = (uint64_t)n_mbs * n_streams;
    if (gid >= (uint64_t)n_mbs && gid < total_mbs) {
        const uint32_t s = (uint32_t)(gid / n_mbs), i = (uint32_t)(gid % n_mbs);
        if (mbs) {
            mpeghip_mb_desc d = mbs[i];
            d.pic += s * n_pics;
            d.coef_off += s * k.coef_units;
            mbs[gid] = d;
        }
        if (xmbs) {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(xmbs + (uint64_t)i * kXDwords);
            u32x4 q0 = src[0], q1 = src[1], q2 = src[2];
            q0.v[1] += s * k.coef_units;
            q0.v[2] += s * 256u;
            q1.v[0] += s * k.frames256;
            q1.v[1] += s * k.frames256;
            q2.v[2] += s * k.rgba256;
            u32x4 *dst = reinterpret_cast<u32x4 *>(xmbs + gid * kXDwords);
            dst[0] = q0;
            dst[1] = q1;
            dst[2] = q2;
            static_assert(kXDwords == 12, "three 16-byte quarters per record");
        }
    }
    const uint64_t total_pics = (uint64_t)n_pics * n_streams;
    if (gid >= (uint64_t)n_pics && gid < total_pics) {
        const uint32_t s = (uint32_t)(gid / n_pics), i = (uint32_t)(gid % n_pics);
        mpeghip_pic_desc p = pics[i];
        p.stream = s;
        p.mb_first += s * n_mbs;
        pics[gid] = p;
    }
}

// Staged submits: rebuild the dense coefficient units from their wire form (video_wire_lane.h).  Grid:
// x = groups of 8 units (4 per workgroup), y = pictures; one wave = 8 units of one picture.
struct WireTab {
    uint32_t unit_first; // first dense unit of the picture in the batch's coefficient array (a multiple of 8)
    uint32_t units;
    uint32_t region;     // dword offset of the picture's wire region: `units` headers, then the payload
    uint32_t reserved;
};
__global__ __launch_bounds__(256) void wire_expand_kernel(const uint32_t *wire, const WireTab *tab, uint32_t pic0,
                                                            uint8_t *coefs)
{
    __shared__ __attribute__((aligned(16))) uint8_t tiles[4 * 1024];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const WireTab t = tab[pic0 + blockIdx.y];
    const uint32_t group = blockIdx.x * 4 + wave;
    if (group * 8 >= t.units)
        return;
    const uint32_t unit = group * 8 + ((uint32_t)lane >> 3);
    const uint32_t *region = wire + t.region;
    WireLane w;
    w.live = unit < t.units;
    w.header = w.live ? region[unit] : 0;
    w.payload = region + t.units;
    uint8_t *tile = tiles + wave * 1024;
    wire_phase_zero(tile, lane);
    wave_lds_handoff();
    wire_phase_scatter(w, tile, lane);
    wave_lds_handoff();
    wire_phase_store(w, tile, lane, coefs + ((uint64_t)t.unit_first + unit) * MPEGHIP_COEF_UNIT);
}

// FNV-1a-64 over Y||Cb||Cr of one slot per stream (mpeg_test.go:221-223); one
// thread per stream — a test aid, not a hot path.
__global__ void hash_kernel(const uint8_t *frames, uint64_t frame_stride, uint32_t slot,
                            uint64_t n_bytes, uint32_t n_streams, uint64_t *out)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams)
        return;
    const uint8_t *p = frames + ((uint64_t)s * MPEGHIP_SLOTS + slot) * frame_stride;
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < n_bytes; i += 8) {
        uint64_t w = *reinterpret_cast<const uint64_t *>(p + i);
        for (int k = 0; k < 8; k++) {
            h ^= (w >> (8 * k)) & 0xff;
            h *= 0x100000001b3ull;
        }
    }
    out[s] = h;
}

template <bool kFma, int kFormat> __global__ __launch_bounds__(kAudioThreads) void audio_kernel(const AudioArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[kAudioLdsFloats];
    const uint32_t stream = blockIdx.x / a.n_chunks, chunk = blockIdx.x % a.n_chunks;
    const int tid = threadIdx.x;
    uint32_t f0, f1;
    audio_chunk_range(a, chunk, f0, f1);
    if (f0 >= f1)
        return; // empty slice (wave-uniform, before any barrier)
    if (a.active && a.active[stream] == 0) { // (workgroup-uniform)
        if (f1 == a.n_frames)
            audio_carry_state(a, stream, tid);
        return;
    }
    const int32_t vpos0 = a.vpos[stream];
    const uint32_t tg0 = f0 * 36, tg1 = f1 * 36, n_steps = (tg1 - tg0 + kStep - 1) / kStep;
    float dreg[16];
    audio_load_window(a, tid, dreg);
    // prologue: samples of step 0 in flight, history from the state or rebuilt
    audio_phase_fetch(a, stream, tg0, tg1, 0, tid, lds);
    if (f0 == 0)
        audio_load_state(a, stream, vpos0, tid, lds);
    else
        audio_phase_warmup(a, stream, f0, tid, lds);
    __syncthreads();
    audio_phase_dct(a, stream, tg0, tg1, 0, tid, lds); // (also puts the samples of step 1 in flight)
    __syncthreads();
    for (uint32_t si = 0; si < n_steps; si++) {
        audio_phase_dct(a, stream, tg0, tg1, si + 1, tid, lds); // wave (si+1)%4; refills the staging buffer for step si+2
        audio_phase_window<kFma, kFormat>(a, stream, vpos0, tg0, tg1, si, tid, dreg, lds);
        __syncthreads();
    }
    if (f1 == a.n_frames) { // the slice that ends the launch owns the state hand-over
        audio_store_state(a, stream, vpos0, tid, lds);
        if (tid == 0)
            audio_store_vpos(a, stream, vpos0);
    }
}

// ================================================================ host side

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? MPEGHIP_ERR_OOM : MPEGHIP_ERR_HIP,             \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct mpeghip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

struct mpeghip_batch {
    mpeghip_video *owner = nullptr;
    mpeghip_pic_desc *d_pics = nullptr;
    mpeghip_mb_desc *d_mbs = nullptr; // filled only for the diagnostic kernels (MPEGHIP_RECON mode < 6)
    uint32_t *d_xmbs = nullptr;       // expanded records of the wave-chunk kernel (mode 6)
    uint32_t *d_wire = nullptr;       // staged submits: coefficient units in wire form (video_wire_lane.h)
    void *d_wtab = nullptr;           //                 one WireTab per picture
    // Submits that come through a pinned staging buffer keep its device image in ONE allocation, filled by one
    // H2D copy: staged submits pictures | WireTab | records | wire regions (d_coefs is separate: the units
    // rebuilt by wire_expand_kernel), plain submits pictures | descriptors | records | coefficients.
    enum Form { Separate, StageBlob, SubmitBlob } form = Separate;
    uint8_t *d_blob = nullptr;
    size_t cap_blob = 0;
    uint8_t *d_coefs = nullptr;
    uint64_t n_pics = 0, n_mbs = 0, coef_bytes = 0;
    uint64_t alg_bytes = 0;
    BlockEntry *d_entries = nullptr; // split path work list, one per coefficient unit
    bool dense_partition = false;   // coefficient units are an ordered partition of the stream: no memset needed
    bool any_rgba = false;
    // host copy of what launch_batch needs to keep the RGBA images in step: per picture of the original
    // (un-replicated) batch {stream, cur, MPEGHIP_PIC_RGBA?, covers every macroblock of the frame?}
    struct PicNote { uint32_t stream; uint8_t cur, rgba, full; };
    std::vector<PicNote> notes;
    uint32_t replicas = 1;
    size_t cap_pics = 0, cap_mbs = 0, cap_xmbs = 0, cap_coefs = 0, cap_entries = 0; // capacities (transient batch reuse)

};

struct mpeghip_video {
    mpeghip_ctx *ctx = nullptr;
    mpeghip_video_info info{};
    uint8_t *d_frames = nullptr;
    uint8_t *d_rgba = nullptr;
    uint8_t *d_qmat = nullptr;    // [n_streams][2][8][16]: per column {8 matrix bytes, 8 premultiplier bytes}
    uint8_t *d_dump = nullptr;    // sink for the static-count stores of the pipelined kernel
    size_t dump_bytes = 0;
    uint64_t *d_hash = nullptr;
    // rgba_sync[stream*3 + slot]: the slot's RGBA image equals the conversion of its planes.  Pictures
    // flagged MPEGHIP_PIC_RGBA convert the macroblocks they write inside the reconstruction kernel; that
    // is the whole story unless a partial picture lands on a slot whose image is out of date — then
    // the whole-frame pass runs as well (launch_batch).
    std::vector<uint8_t> rgba_sync;
    // mpeghip_video_submit: two descriptor batches with pinned host staging, used alternately, so
    // that the caller can parse picture N+1 while picture N's copy and kernel are in flight
    struct Staging {
        mpeghip_batch batch;
        uint8_t *h = nullptr;      // pinned
        size_t cap_h = 0;
        hipEvent_t done = nullptr; // recorded after the batch's kernel
        bool in_flight = false;
    } staging[2];
    int next_staging = 0;
    struct mpeghip_stage *stage = nullptr; // the open mpeghip_video_stage_begin, if any
    uint8_t *bounce = nullptr;             // pinned: read_planes / read_rgba land here first
    size_t bounce_cap = 0;
};

// mpeghip_video_stage_*: one submit assembled in a staging buffer by several host threads
struct mpeghip_stage {
    mpeghip_video *v = nullptr;
    mpeghip_video::Staging *sg = nullptr;
    uint32_t n_pics = 0, n_mbs = 0;
    uint64_t coef_units = 0;
    std::vector<uint32_t> mb_first, mb_count;   // per picture: its records [mb_first, mb_first + mb_count)
    std::vector<uint64_t> unit_first, units;    // per picture: its coefficient units (unit_first: multiples of 8)
    std::vector<uint64_t> alg;                  // per picture, written by its put
    std::vector<uint8_t> done;                  // per picture: put succeeded
    size_t x_at = 0, tab_at = 0, wire0 = 0;     // staging layout: pictures | WireTab | records | wire regions
    uint64_t wire_cap_dwords = 0;               // room for all regions if every unit travelled dense
    std::atomic<uint64_t> wire_used{0};         // dwords handed out so far: a put packs its picture into scratch
                                                // memory of its thread, then takes exactly the room it needs, so
                                                // that the regions form one contiguous block = one H2D copy
    std::atomic<int> error{MPEGHIP_OK};         // first failed put
    std::mutex error_lock;
    std::string error_text;
};

struct mpeghip_audio {
    mpeghip_ctx *ctx = nullptr;
    uint32_t n_streams = 0;
    int fma = 0;
    float *d_ring = nullptr, *d_ring_alt = nullptr;   // state before / after a launch (swapped each launch)
    int32_t *d_vpos = nullptr, *d_vpos_alt = nullptr;
    float *d_window = nullptr;
    int32_t *d_samples = nullptr;
    void *d_out = nullptr;
    size_t cap_samples = 0, cap_out = 0;
    uint8_t *d_active = nullptr;   // [n_streams] mask of mpeghip_audio_synth_masked
};

static const uint8_t k_default_intra[64] = { // ISO 11172-2 default intra matrix (video.go:1055-1064)
    8,  16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37,
    19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
    22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58,
    26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83};

// AAN-style premultiplier (video.go:1077-1086): round(32 * s_r * s_c) style scale
// factors of the reference's IDCT; symmetric, so row- and column-major coincide.
static const uint8_t k_premult[64] = {
    32, 44, 42, 38, 32, 25, 17, 9,  44, 62, 58, 52, 44, 35, 24, 12,
    42, 58, 55, 49, 42, 33, 23, 12, 38, 52, 49, 44, 38, 30, 20, 10,
    32, 44, 42, 38, 32, 25, 17, 9,  25, 35, 33, 30, 25, 20, 14, 7,
    17, 24, 23, 20, 17, 14, 9,  5,  9,  12, 12, 10, 9,  7,  5,  2};

// One stream's device table: for each class (intra, non-intra) and column c, the 8
// matrix entries of that column (rows 0..7) followed by the 8 premultipliers.
static void make_qtable(uint8_t out[256], const uint8_t intra[64], const uint8_t non_intra[64])
{
    for (int cls = 0; cls < 2; cls++)
        for (int c = 0; c < 8; c++)
            for (int r = 0; r < 8; r++) {
                out[cls * 128 + c * 16 + r] = (cls ? non_intra : intra)[r * 8 + c];
                out[cls * 128 + c * 16 + 8 + r] = k_premult[r * 8 + c];
            }
}

extern "C" {

int mpeghip_abi_version(void) { return MPEGHIP_ABI_VERSION; }
const char *mpeghip_last_error(void) { return g_err; }

int mpeghip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
        return fail(MPEGHIP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int mpeghip_ctx_create(int device, void *stream, mpeghip_ctx **out)
{
    if (!out)
        return fail(MPEGHIP_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MPEGHIP_ERR_NO_DEVICE, "no HIP device (%s); libmpeghip has no CPU path",
                    e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fail(MPEGHIP_ERR_NO_DEVICE, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MPEGHIP_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device,
                    prop.gcnArchName);
    mpeghip_ctx *c = new (std::nothrow) mpeghip_ctx();
    if (!c)
        return fail(MPEGHIP_ERR_OOM, "host allocation failed");
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete c;
            return fail(MPEGHIP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(se));
        }
        c->owns_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return fail(MPEGHIP_ERR_HIP, "hipEventCreate failed");
    }
    *out = c;
    return MPEGHIP_OK;
}

void mpeghip_ctx_destroy(mpeghip_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->ev0)
        (void)hipEventDestroy(c->ev0);
    if (c->ev1)
        (void)hipEventDestroy(c->ev1);
    if (c->owns_stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
}

int mpeghip_ctx_sync(mpeghip_ctx *c)
{
    if (!c)
        return fail(MPEGHIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPEGHIP_OK;
}

void *mpeghip_pinned_alloc(mpeghip_ctx *c, size_t bytes)
{
    if (!c)
        return nullptr;
    void *p = nullptr;
    (void)hipSetDevice(c->device);
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        fail(MPEGHIP_ERR_OOM, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

void mpeghip_pinned_free(mpeghip_ctx *c, void *p)
{
    if (c && p) {
        (void)hipSetDevice(c->device);
        (void)hipHostFree(p);
    }
}

int mpeghip_timer_start(mpeghip_ctx *c)
{
    if (!c)
        return fail(MPEGHIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return MPEGHIP_OK;
}

int mpeghip_timer_stop_ms(mpeghip_ctx *c, float *ms)
{
    if (!c || !ms)
        return fail(MPEGHIP_ERR_INVALID, "NULL argument");
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return MPEGHIP_OK;
}

// -------------------------------------------------------------------- video

static uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

int mpeghip_video_open(mpeghip_ctx *c, uint32_t width, uint32_t height, uint32_t n_streams, mpeghip_video **out)
{
    if (!c || !out)
        return fail(MPEGHIP_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (width == 0 || height == 0 || width > 4095 || height > 4095 || n_streams == 0)
        return fail(MPEGHIP_ERR_INVALID, "bad geometry %ux%u x %u streams", width, height, n_streams);
    HIP_TRY(hipSetDevice(c->device));
    mpeghip_video *v = new (std::nothrow) mpeghip_video();
    if (!v)
        return fail(MPEGHIP_ERR_OOM, "host allocation failed");
    v->ctx = c;
    mpeghip_video_info &in = v->info;
    in.width = width;
    in.height = height;
    in.mb_w = (width + 15) >> 4; // video.go:314-322
    in.mb_h = (height + 15) >> 4;
    in.luma_w = in.mb_w << 4;
    in.luma_h = in.mb_h << 4;
    in.chroma_w = in.mb_w << 3;
    in.chroma_h = in.mb_h << 3;
    in.n_streams = n_streams;
    in.luma_bytes = (uint64_t)in.luma_w * in.luma_h;
    in.chroma_bytes = (uint64_t)in.chroma_w * in.chroma_h;
    in.frame_bytes = in.luma_bytes + 2 * in.chroma_bytes + (uint64_t)in.luma_w * 16; // video.go:340
    in.frame_stride = align_up(in.frame_bytes + 64, 256); // slack keeps 8-byte row loads inside the slot
    in.rgba_bytes = (uint64_t)width * height * 4;
    const uint64_t total = in.frame_stride * MPEGHIP_SLOTS * n_streams;
    int rc = MPEGHIP_OK;
    do {
        if (hipMalloc((void **)&v->d_frames, total) != hipSuccess) {
            rc = fail(MPEGHIP_ERR_OOM, "hipMalloc(%llu) for the frame store failed", (unsigned long long)total);
            break;
        }
        int n_cu = 256;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
        v->dump_bytes = (size_t)(n_cu + 8) * 32 /* waves per CU */ * 512;
        if (hipMalloc((void **)&v->d_qmat, (size_t)n_streams * 256) != hipSuccess ||
            hipMalloc((void **)&v->d_dump, v->dump_bytes) != hipSuccess ||
            hipMalloc((void **)&v->d_hash, (size_t)n_streams * 8) != hipSuccess) {
            rc = fail(MPEGHIP_ERR_OOM, "hipMalloc for tables failed");
            break;
        }
        if (hipMemsetAsync(v->d_frames, 0, total, c->stream) != hipSuccess) {
            rc = fail(MPEGHIP_ERR_HIP, "hipMemsetAsync failed");
            break;
        }
        std::vector<uint8_t> qm((size_t)n_streams * 256);
        uint8_t non_intra[64];
        memset(non_intra, 16, 64); // video.go:1066-1075
        for (uint32_t s = 0; s < n_streams; s++)
            make_qtable(&qm[(size_t)s * 256], k_default_intra, non_intra);
        if (hipMemcpy(v->d_qmat, qm.data(), qm.size(), hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(MPEGHIP_ERR_HIP, "table upload failed");
            break;
        }
        if (hipStreamSynchronize(c->stream) != hipSuccess) {
            rc = fail(MPEGHIP_ERR_HIP, "sync failed");
            break;
        }
    } while (0);
    if (rc != MPEGHIP_OK) {
        mpeghip_video_close(v);
        return rc;
    }
    v->staging[0].batch.owner = v->staging[1].batch.owner = v;
    v->rgba_sync.assign((size_t)n_streams * MPEGHIP_SLOTS, 0);
    *out = v;
    return MPEGHIP_OK;
}

// The descriptor arrays are either allocations of their own (resident batches) or parts of d_blob; a staging
// batch may change from one blob form to the other between submits.
static void batch_drop_descriptors(mpeghip_batch *b)
{
    if (b->form == mpeghip_batch::Separate) {
        if (b->d_pics)
            (void)hipFree(b->d_pics);
        if (b->d_xmbs)
            (void)hipFree(b->d_xmbs);
    } else {
        if (b->d_blob)
            (void)hipFree(b->d_blob);
        if (b->form == mpeghip_batch::SubmitBlob) { // these were parts of the blob as well
            b->d_mbs = nullptr;
            b->d_coefs = nullptr;
            b->cap_mbs = b->cap_coefs = 0;
        }
    }
    b->form = mpeghip_batch::Separate;
    b->d_blob = nullptr;
    b->cap_blob = 0;
    b->d_pics = nullptr;
    b->d_xmbs = nullptr;
    b->d_wire = nullptr;
    b->d_wtab = nullptr;
    b->cap_pics = b->cap_xmbs = 0;
}

static void batch_release(mpeghip_batch *b)
{
    batch_drop_descriptors(b);
    if (b->d_mbs)
        (void)hipFree(b->d_mbs);
    if (b->d_coefs)
        (void)hipFree(b->d_coefs);
    if (b->d_entries)
        (void)hipFree(b->d_entries);
    b->d_entries = nullptr;
    b->cap_entries = 0;
    b->d_pics = nullptr;
    b->d_mbs = nullptr;
    b->d_xmbs = nullptr;
    b->d_wire = nullptr;
    b->d_wtab = nullptr;
    b->d_coefs = nullptr;
    b->cap_pics = b->cap_mbs = b->cap_coefs = 0;
}

void mpeghip_video_close(mpeghip_video *v)
{
    if (!v)
        return;
    (void)hipSetDevice(v->ctx->device);
    (void)hipStreamSynchronize(v->ctx->stream);
    delete v->stage; // a stage that was begun and never committed
    v->stage = nullptr;
    if (v->bounce)
        (void)hipHostFree(v->bounce);
    for (auto &sg : v->staging) {
        batch_release(&sg.batch);
        if (sg.h)
            (void)hipHostFree(sg.h);
        if (sg.done)
            (void)hipEventDestroy(sg.done);
    }
    if (v->d_frames)
        (void)hipFree(v->d_frames);
    if (v->d_rgba)
        (void)hipFree(v->d_rgba);
    if (v->d_qmat)
        (void)hipFree(v->d_qmat);
    if (v->d_dump)
        (void)hipFree(v->d_dump);
    if (v->d_hash)
        (void)hipFree(v->d_hash);
    delete v;
}

int mpeghip_video_info_get(const mpeghip_video *v, mpeghip_video_info *info)
{
    if (!v || !info)
        return fail(MPEGHIP_ERR_INVALID, "NULL argument");
    *info = v->info;
    return MPEGHIP_OK;
}

int mpeghip_video_set_quant(mpeghip_video *v, uint32_t stream, const uint8_t intra[64], const uint8_t non_intra[64])
{
    if (!v || !intra || !non_intra || stream >= v->info.n_streams)
        return fail(MPEGHIP_ERR_INVALID, "bad argument");
    uint8_t t[256];
    make_qtable(t, intra, non_intra);
    HIP_TRY(hipSetDevice(v->ctx->device));
    HIP_TRY(hipStreamSynchronize(v->ctx->stream)); // earlier pictures may still read the old matrices
    HIP_TRY(hipMemcpy(v->d_qmat + (size_t)stream * 256, t, 256, hipMemcpyHostToDevice));
    return MPEGHIP_OK;
}

static int ensure_rgba(mpeghip_video *v)
{
    if (v->d_rgba)
        return MPEGHIP_OK;
    const uint64_t total = align_up(v->info.rgba_bytes, 256) * MPEGHIP_SLOTS * v->info.n_streams;
    HIP_TRY(hipMalloc((void **)&v->d_rgba, total));
    // RGBA of an all-zero frame is (0,135,0,255), not zero: convert the (zero or
    // already decoded) planes so image and planes agree from the start.
    for (uint32_t slot = 0; slot < MPEGHIP_SLOTS; slot++) {
        int rc = mpeghip_video_rgba_convert(v, slot, 0, v->info.n_streams);
        if (rc != MPEGHIP_OK)
            return rc;
    }
    return MPEGHIP_OK;
}

static bool wants_rgba(const mpeghip_pic_desc *pics, uint32_t n_pics);
static uint64_t rgba_stride_of(const mpeghip_video *v) { return align_up(v->info.rgba_bytes, 256); }

// Host-side validation of one submit; also totals the algorithmic bytes.
static int validate_pic(const mpeghip_video_info &in, const mpeghip_pic_desc &pd, uint32_t p)
{
    if (pd.stream >= in.n_streams || pd.cur >= MPEGHIP_SLOTS || pd.fwd >= MPEGHIP_SLOTS || pd.bwd >= MPEGHIP_SLOTS)
        return fail(MPEGHIP_ERR_INVALID, "picture %u: bad stream/slot", p);
    return MPEGHIP_OK;
}

// One macroblock of picture `pd`: field ranges, coefficient extent inside [0, coef_units), prediction
// reads inside the frame buffer.  *units = coefficient units it owns; adds its algorithmic bytes to *alg.
static int validate_mb(const mpeghip_video_info &in, const mpeghip_pic_desc &pd, const mpeghip_mb_desc &m, uint32_t i,
                       uint64_t coef_units, uint64_t *units_out, uint64_t *alg)
{
    if (m.mb_x >= in.mb_w || m.mb_y >= in.mb_h)
        return fail(MPEGHIP_ERR_INVALID, "macroblock %u: position (%u,%u) outside %ux%u", i, m.mb_x, m.mb_y, in.mb_w,
                    in.mb_h);
    const bool intra = m.flags & MPEGHIP_MB_INTRA;
    const uint32_t nref = ((m.flags & MPEGHIP_MB_REF_FWD) ? 1 : 0) + ((m.flags & MPEGHIP_MB_REF_BWD) ? 1 : 0);
    if ((intra && nref != 0) || (!intra && nref != 1))
        return fail(MPEGHIP_ERR_INVALID, "macroblock %u: flags 0x%x name %u references", i, m.flags, nref);
    if (m.cbp > 0x3f)
        return fail(MPEGHIP_ERR_INVALID, "macroblock %u: cbp 0x%x", i, m.cbp);
    const uint32_t nb = (uint32_t)__builtin_popcount(m.cbp);
    const bool raw = m.flags & MPEGHIP_MB_COEF_RAW;
    const uint64_t units = (uint64_t)nb * (raw ? 2 : 1);
    if (nb && (uint64_t)m.coef_off + units > coef_units)
        return fail(MPEGHIP_ERR_INVALID, "macroblock %u: coefficient blocks beyond the buffer", i);
    if (!raw && nb && (m.qscale == 0 || m.qscale > 31))
        return fail(MPEGHIP_ERR_INVALID, "macroblock %u: quantiser_scale %u", i, m.qscale);
    uint64_t ref_bytes = 0;
    if (!intra) {
        // extents of the reference's copyBlock reads (video_noasm.go:48-80): Go
        // indexes src[:cap(src)], i.e. [plane start, end of base); anything else panics.
        const int64_t cap_y = (int64_t)in.frame_bytes;
        const int64_t cap_c0 = (int64_t)(in.frame_bytes - in.luma_bytes);
        const int64_t cap_c1 = (int64_t)(in.frame_bytes - in.luma_bytes - in.chroma_bytes);
        const int mh = m.mv_x, mv = m.mv_y;
        const int64_t lsi = ((int64_t)(m.mb_y << 4) + (mv >> 1)) * in.luma_w + (m.mb_x << 4) + (mh >> 1);
        const int loh = mh & 1, lov = mv & 1;
        const int64_t llast = lsi + (int64_t)(15 + lov) * in.luma_w + 15 + loh;
        const int cmh = mh / 2, cmv = mv / 2;
        const int64_t csi = ((int64_t)(m.mb_y << 3) + (cmv >> 1)) * in.chroma_w + (m.mb_x << 3) + (cmh >> 1);
        const int coh = cmh & 1, cov = cmv & 1;
        const int64_t clast = csi + (int64_t)(7 + cov) * in.chroma_w + 7 + coh;
        if (lsi < 0 || llast >= cap_y || csi < 0 || clast >= cap_c1 || clast >= cap_c0)
            return fail(MPEGHIP_ERR_RANGE, "macroblock %u at (%u,%u): motion vector (%d,%d) reads outside the frame buffer",
                        i, m.mb_x, m.mb_y, mh, mv);
        ref_bytes = (uint64_t)(16 + lov) * (16 + loh) + 2ull * (8 + cov) * (8 + coh);
    }
    *units_out = units;
    *alg += 32 + units * MPEGHIP_COEF_UNIT + ref_bytes + (intra ? 64ull * nb : 384);
    if (pd.flags & MPEGHIP_PIC_RGBA)
        *alg += 1024;
    return MPEGHIP_OK;
}

static XGeom record_geometry(const mpeghip_video *v)
{
    XGeom geom;
    geom.luma_w = v->info.luma_w;
    geom.chroma_w = v->info.chroma_w;
    geom.frame_stride = v->info.frame_stride;
    geom.rgba_stride = rgba_stride_of(v);
    return geom;
}

static int validate(const mpeghip_video *v, const mpeghip_pic_desc *pics, uint32_t n_pics,
                    const mpeghip_mb_desc *mbs, uint32_t n_mbs, size_t coef_bytes, uint64_t *alg_bytes,
                    bool *dense_partition, uint32_t *xrec = nullptr)
{
    // xrec != NULL: also write each macroblock's expanded record (video_compact_lane.h: expand_mb) — the same
    // pass has just checked every field the record is computed from
    const mpeghip_video_info &in = v->info;
    const XGeom geom = record_geometry(v);
    std::vector<XPic> xpics;
    if (n_pics && !pics)
        return fail(MPEGHIP_ERR_INVALID, "pics is NULL");
    if (n_mbs && !mbs)
        return fail(MPEGHIP_ERR_INVALID, "mbs is NULL");
    if (coef_bytes % MPEGHIP_COEF_UNIT)
        return fail(MPEGHIP_ERR_INVALID, "coef_bytes %zu is not a multiple of 128", coef_bytes);
    for (uint32_t p = 0; p < n_pics; p++) {
        const int rc = validate_pic(in, pics[p], p);
        if (rc != MPEGHIP_OK)
            return rc;
        if ((uint64_t)pics[p].mb_first + pics[p].mb_count > n_mbs)
            return fail(MPEGHIP_ERR_INVALID, "picture %u: macroblock range out of bounds", p);
    }
    if (xrec) {
        xpics.resize(n_pics);
        for (uint32_t p = 0; p < n_pics; p++)
            xpics[p] = expand_pic(geom, pics[p]);
    }
    uint64_t alg = 0;
    const uint64_t coef_units = coef_bytes / MPEGHIP_COEF_UNIT;
    uint64_t next_unit = 0;
    bool dense = true;
    for (uint32_t i = 0; i < n_mbs; i++) {
        const mpeghip_mb_desc &m = mbs[i];
        if (m.pic >= n_pics)
            return fail(MPEGHIP_ERR_INVALID, "macroblock %u: picture index %u out of range", i, m.pic);
        uint64_t units = 0;
        const int rc = validate_mb(in, pics[m.pic], m, i, coef_units, &units, &alg);
        if (rc != MPEGHIP_OK)
            return rc;
        if (units) {
            if (m.coef_off != next_unit)
                dense = false;
            next_unit = (uint64_t)m.coef_off + units;
        }
        if (xrec)
            expand_mb(geom, xpics[m.pic], m, xrec + (size_t)i * kXDwords);
    }
    if (xrec)
        mark_chunk_runs(xrec, n_mbs);
    if (alg_bytes)
        *alg_bytes = alg;
    if (dense_partition)
        *dense_partition = dense && next_unit == coef_units;
    return MPEGHIP_OK;
}

static int grow(void **p, size_t *cap, size_t need)
{
    if (need <= *cap)
        return MPEGHIP_OK;
    if (*p)
        (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    size_t want = need + need / 4 + 4096;
    if (hipMalloc(p, want) != hipSuccess)
        return fail(MPEGHIP_ERR_OOM, "hipMalloc(%zu) failed", want);
    *cap = want;
    return MPEGHIP_OK;
}

// Development knob (not part of the ABI): MPEGHIP_RECON="mode,waves,blocks_per_cu".
//   mode 6 (default): wave-chunk kernel: one wave = 4 macroblocks, dense residual stage, no barrier
//                     ("waves" = 4 -> 4 waves/block, 8 -> 8, 16 -> 2)
//   mode 5: compact fused kernel (dense residual stage inside the workgroup) (+ RGBA pass)
//   mode 4: split path, K1 prediction + K2 dense residual (+ RGBA pass)
//   mode 0: fused one-wave-per-macroblock kernel;  1-3: its persistent / pipelined variants
struct ReconKnob {
    int mode = 6, waves = 4, bpc = 4;
};
static ReconKnob recon_knob()
{
    ReconKnob r;
    if (const char *e = getenv("MPEGHIP_RECON"))
        sscanf(e, "%d,%d,%d", &r.mode, &r.waves, &r.bpc);
    if (r.waves != 4 && r.waves != 8 && r.waves != 16)
        r.waves = 8;
    if (r.mode < 0 || r.mode > 6)
        r.mode = 6;
    return r;
}

static int launch_batch(mpeghip_video *v, const mpeghip_batch *b)
{
    if (b->n_mbs == 0)
        return MPEGHIP_OK;
    const mpeghip_video_info &in = v->info;
    VideoArgs a;
    a.frames = v->d_frames;
    a.frame_stride = in.frame_stride;
    a.luma_w = in.luma_w;
    a.luma_h = in.luma_h;
    a.chroma_w = in.chroma_w;
    a.chroma_h = in.chroma_h;
    a.luma_bytes = (uint32_t)in.luma_bytes;
    a.chroma_bytes = (uint32_t)in.chroma_bytes;
    a.pics = b->d_pics;
    a.mbs = b->d_mbs;
    a.xmbs = b->d_xmbs;
    a.coefs = b->d_coefs;
    a.qmat = v->d_qmat;
    a.dump = v->d_dump;
    a.n_mbs = (uint32_t)b->n_mbs;
    a.width = in.width;
    a.height = in.height;
    a.rgba = v->d_rgba;
    a.rgba_stride = rgba_stride_of(v);
    const ReconKnob knob = recon_knob();
    const int mode = knob.mode, waves = knob.waves, bpc = knob.bpc;
    if (mode == 6) {
        hipStream_t st = v->ctx->stream;
        const uint32_t n_chunks = (uint32_t)((b->n_mbs + kWcMbs - 1) / kWcMbs);
#define LAUNCH_WC(W)                                                                                                   \
    do {                                                                                                               \
        if (b->any_rgba)                                                                                               \
            hipLaunchKernelGGL((recon_wc_kernel<W, true>), dim3((n_chunks + W - 1) / W), dim3(W * 64), 0, st, a, n_chunks);  \
        else                                                                                                           \
            hipLaunchKernelGGL((recon_wc_kernel<W, false>), dim3((n_chunks + W - 1) / W), dim3(W * 64), 0, st, a, n_chunks); \
    } while (0)
        if (waves == 8)
            LAUNCH_WC(8);
        else if (waves == 16)
            LAUNCH_WC(2);
        else
            LAUNCH_WC(4);
#undef LAUNCH_WC
        HIP_TRY(hipGetLastError());
        // Frame.RGBA bookkeeping: the kernel has converted every macroblock that flagged pictures wrote.
        // A whole-frame pass is still owed when a flagged picture covered only part of a frame whose
        // image was out of date (an unflagged picture or write_planes touched the slot since).
        bool whole_frames = false;
        for (uint32_t r = 0; r < b->replicas; r++)
            for (const mpeghip_batch::PicNote &n : b->notes) {
                uint8_t &sync = v->rgba_sync[((size_t)n.stream + r) * MPEGHIP_SLOTS + n.cur];
                if (!n.rgba)
                    sync = 0;
                else if (n.full)
                    sync = 1;
                else if (!sync)
                    whole_frames = true, sync = 1;
            }
        if (whole_frames) {
            const uint32_t quads = (in.width + 3) / 4;
            for (uint64_t p0 = 0; p0 < b->n_pics; p0 += 32768) {
                const uint32_t np = (uint32_t)(b->n_pics - p0 < 32768 ? b->n_pics - p0 : 32768);
                hipLaunchKernelGGL(rgba_pics_kernel, dim3((quads + 63) / 64, (in.height + 7) / 8, np), dim3(256), 0, st, a,
                                   (uint32_t)p0);
            }
            HIP_TRY(hipGetLastError());
        }
        return MPEGHIP_OK;
    }
    if (mode == 5) {
        hipStream_t st = v->ctx->stream;
        const uint32_t blocks = (uint32_t)((b->n_mbs + kChunkMbs - 1) / kChunkMbs);
        hipLaunchKernelGGL(recon_compact_kernel, dim3(blocks), dim3(kChunkMbs * 64), 0, st, a);
        HIP_TRY(hipGetLastError());
        if (b->any_rgba) {
            const uint32_t quads = (in.width + 3) / 4;
            for (uint64_t p0 = 0; p0 < b->n_pics; p0 += 32768) {
                const uint32_t np = (uint32_t)(b->n_pics - p0 < 32768 ? b->n_pics - p0 : 32768);
                hipLaunchKernelGGL(rgba_pics_kernel, dim3((quads + 63) / 64, (in.height + 7) / 8, np), dim3(256), 0, st, a,
                                   (uint32_t)p0);
            }
            HIP_TRY(hipGetLastError());
        }
        return MPEGHIP_OK;
    }
    if (mode == 4) {
        hipStream_t st = v->ctx->stream;
        SplitArgs s;
        s.v = a;
        s.entries = b->d_entries;
        s.n_units = (uint32_t)(b->coef_bytes / MPEGHIP_COEF_UNIT);
        if (!b->dense_partition && s.n_units)
            HIP_TRY(hipMemsetAsync(b->d_entries, 0xff, (size_t)s.n_units * sizeof(BlockEntry), st));
        const uint32_t pred_blocks = (uint32_t)((b->n_mbs + 2 * waves - 1) / (2 * waves));
        const uint32_t resid_blocks = (s.n_units + 8 * waves - 1) / (8 * waves);
#define LAUNCH_SPLIT(W)                                                                                                \
    do {                                                                                                               \
        hipLaunchKernelGGL((pred_kernel<W>), dim3(pred_blocks), dim3(W * 64), 0, st, s);                               \
        if (resid_blocks)                                                                                              \
            hipLaunchKernelGGL((resid_kernel<W>), dim3(resid_blocks), dim3(W * 64), 0, st, s);                         \
    } while (0)
        if (waves == 4)
            LAUNCH_SPLIT(4);
        else if (waves == 16)
            LAUNCH_SPLIT(16);
        else
            LAUNCH_SPLIT(8);
#undef LAUNCH_SPLIT
        HIP_TRY(hipGetLastError());
        if (b->any_rgba) {
            const uint32_t quads = (in.width + 3) / 4;
            for (uint64_t p0 = 0; p0 < b->n_pics; p0 += 32768) {
                const uint32_t np = (uint32_t)(b->n_pics - p0 < 32768 ? b->n_pics - p0 : 32768);
                hipLaunchKernelGGL(rgba_pics_kernel, dim3((quads + 63) / 64, (in.height + 7) / 8, np), dim3(256), 0, st, a,
                                   (uint32_t)p0);
            }
            HIP_TRY(hipGetLastError());
        }
        return MPEGHIP_OK;
    }
    const uint32_t n_chunks = (uint32_t)((b->n_mbs + waves - 1) / waves);
    uint32_t blocks = n_chunks;
    if (mode != 0) {
        int n_cu = 256;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, v->ctx->device);
        uint32_t want = (uint32_t)n_cu * (uint32_t)(bpc < 1 ? 1 : bpc);
        want = (want + 7) / 8 * 8;
        while ((size_t)want * (size_t)waves * 512 > v->dump_bytes && want > 8)
            want -= 8;
        blocks = want;
    }
    hipStream_t st = v->ctx->stream;
#define LAUNCH(W, M) hipLaunchKernelGGL((recon_kernel<W, M>), dim3(blocks), dim3(W * 64), 0, st, a, n_chunks)
#define LAUNCH_W(W)                                                                                                    \
    do {                                                                                                               \
        if (mode == 0)                                                                                                 \
            LAUNCH(W, 0);                                                                                              \
        else if (mode == 1)                                                                                            \
            LAUNCH(W, 1);                                                                                              \
        else if (mode == 3)                                                                                            \
            LAUNCH(W, 3);                                                                                              \
        else                                                                                                           \
            LAUNCH(W, 2);                                                                                              \
    } while (0)
    if (waves == 4)
        LAUNCH_W(4);
    else if (waves == 16)
        LAUNCH_W(16);
    else
        LAUNCH_W(8);
#undef LAUNCH_W
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    return MPEGHIP_OK;
}

static bool wants_rgba(const mpeghip_pic_desc *pics, uint32_t n_pics)
{
    for (uint32_t p = 0; p < n_pics; p++)
        if (pics[p].flags & MPEGHIP_PIC_RGBA)
            return true;
    return false;
}

// `sg` != nullptr: b is that staging slot's batch; the host arrays are copied into its pinned buffer
// and the call returns with the copies still in flight.  Otherwise (resident batches) the copies
// read the caller's pageable memory and the call waits for them.
static int upload_into(mpeghip_video *v, mpeghip_batch *b, const mpeghip_pic_desc *pics, uint32_t n_pics,
                       const mpeghip_mb_desc *mbs, uint32_t n_mbs, const void *coefs, size_t coef_bytes,
                       uint32_t replicas, mpeghip_video::Staging *sg = nullptr)
{
    if (n_mbs > 0 && coef_bytes > 0 && !coefs)
        return fail(MPEGHIP_ERR_INVALID, "coefs is NULL");
    if ((uint64_t)n_mbs * replicas > 0xffffffffull || (uint64_t)(coef_bytes / MPEGHIP_COEF_UNIT) * replicas > 0xffffffffull)
        return fail(MPEGHIP_ERR_INVALID, "batch too large for 32-bit descriptor indices");
    HIP_TRY(hipSetDevice(v->ctx->device));
    // The wave-chunk kernel (mode 6) reads expanded records, which the validation pass below writes straight
    // into the buffer the H2D copy reads; the ABI descriptors go to the device only for the diagnostic
    // kernels.  A submit carries one of the two; a resident batch both (it may be run under either knob).
    const bool records = !sg || recon_knob().mode == 6, descs = !sg || !records;
    const size_t pb = sizeof(mpeghip_pic_desc) * (size_t)n_pics;
    const size_t mb = descs ? sizeof(mpeghip_mb_desc) * (size_t)n_mbs : 0;
    const size_t xb = records ? sizeof(uint32_t) * kXDwords * (size_t)n_mbs : 0;
    const size_t x_at = (pb + mb + 63) & ~(size_t)63, coef_at = (x_at + xb + 63) & ~(size_t)63;
    std::vector<uint32_t> xhost; // (resident batches: pageable, this call waits for the copies anyway)
    uint32_t *xrec = nullptr;
    hipStream_t st = v->ctx->stream;
    if (sg) {
        if (sg->in_flight) { // two submits ago: normally long finished
            HIP_TRY(hipEventSynchronize(sg->done));
            sg->in_flight = false;
        }
        const size_t need = coef_at + coef_bytes + 64;
        if (need > sg->cap_h) {
            if (sg->h)
                (void)hipHostFree(sg->h);
            sg->h = nullptr;
            sg->cap_h = 0;
            const size_t cap = need + need / 2;
            HIP_TRY(hipHostMalloc((void **)&sg->h, cap, hipHostMallocDefault));
            sg->cap_h = cap;
        }
        if (!sg->done)
            HIP_TRY(hipEventCreateWithFlags(&sg->done, hipEventDisableTiming));
        if (records)
            xrec = reinterpret_cast<uint32_t *>(sg->h + x_at);
    } else if (records) {
        xhost.resize((size_t)n_mbs * kXDwords);
        xrec = xhost.data();
    }
    int rc = validate(v, pics, n_pics, mbs, n_mbs, coef_bytes, &b->alg_bytes, &b->dense_partition, xrec);
    b->any_rgba = wants_rgba(pics, n_pics);
    if (rc != MPEGHIP_OK)
        return rc;
    if (b->any_rgba) {
        rc = ensure_rgba(v);
        if (rc != MPEGHIP_OK)
            return rc;
    }
    if (sg) {
        if (pb)
            memcpy(sg->h, pics, pb);
        if (mb)
            memcpy(sg->h + pb, mbs, mb);
        if (coef_bytes)
            memcpy(sg->h + coef_at, coefs, coef_bytes);
        pics = reinterpret_cast<const mpeghip_pic_desc *>(sg->h);
        mbs = reinterpret_cast<const mpeghip_mb_desc *>(sg->h + pb);
        coefs = sg->h + coef_at;
    }
    if (sg) {
        // one allocation = the image of the staging buffer, one copy
        if (b->form != mpeghip_batch::SubmitBlob) {
            batch_drop_descriptors(b);
            if (b->d_mbs) // allocations of their own so far; parts of the blob from now on
                (void)hipFree(b->d_mbs);
            if (b->d_coefs)
                (void)hipFree(b->d_coefs);
            b->d_mbs = nullptr;
            b->d_coefs = nullptr;
            b->cap_mbs = b->cap_coefs = 0;
        }
        b->form = mpeghip_batch::SubmitBlob;
        const size_t total = coef_at + coef_bytes;
        if ((rc = grow((void **)&b->d_blob, &b->cap_blob, total + 256)) != 0 ||
            (rc = grow((void **)&b->d_entries, &b->cap_entries, (coef_bytes / MPEGHIP_COEF_UNIT) * sizeof(BlockEntry) + 64)) != 0)
            return rc;
        b->d_pics = reinterpret_cast<mpeghip_pic_desc *>(b->d_blob);
        b->d_mbs = reinterpret_cast<mpeghip_mb_desc *>(b->d_blob + pb);
        b->d_xmbs = reinterpret_cast<uint32_t *>(b->d_blob + x_at);
        b->d_coefs = b->d_blob + coef_at;
        HIP_TRY(hipMemcpyAsync(b->d_blob, sg->h, total, hipMemcpyHostToDevice, st));
    } else {
        if (b->form != mpeghip_batch::Separate)
            batch_drop_descriptors(b);
        if ((rc = grow((void **)&b->d_pics, &b->cap_pics, sizeof(mpeghip_pic_desc) * (size_t)n_pics * replicas + 16)) != 0 ||
            (rc = grow((void **)&b->d_mbs, &b->cap_mbs, mb * replicas + 32)) != 0 ||
            (rc = grow((void **)&b->d_xmbs, &b->cap_xmbs, xb * replicas + 64 * kWcMbs)) != 0 ||
            (rc = grow((void **)&b->d_coefs, &b->cap_coefs, coef_bytes * replicas + 256)) != 0 ||
            (rc = grow((void **)&b->d_entries, &b->cap_entries, (coef_bytes / MPEGHIP_COEF_UNIT) * replicas * sizeof(BlockEntry) + 64)) != 0)
            return rc;
        if (pb)
            HIP_TRY(hipMemcpyAsync(b->d_pics, pics, pb, hipMemcpyHostToDevice, st));
        if (mb)
            HIP_TRY(hipMemcpyAsync(b->d_mbs, mbs, mb, hipMemcpyHostToDevice, st));
        if (xb)
            HIP_TRY(hipMemcpyAsync(b->d_xmbs, xrec, xb, hipMemcpyHostToDevice, st));
        if (coef_bytes)
            HIP_TRY(hipMemcpyAsync(b->d_coefs, coefs, coef_bytes, hipMemcpyHostToDevice, st));
    }
    if (replicas > 1) {
        for (uint32_t s = 1; s < replicas && coef_bytes; s++)
            HIP_TRY(hipMemcpyAsync(b->d_coefs + (size_t)s * coef_bytes, b->d_coefs, coef_bytes, hipMemcpyDeviceToDevice, st));
        const uint64_t work = (uint64_t)(n_mbs > n_pics ? n_mbs : n_pics) * replicas;
        ReplicateSteps k;
        k.coef_units = (uint32_t)(coef_bytes / MPEGHIP_COEF_UNIT);
        k.frames256 = (uint32_t)((MPEGHIP_SLOTS * v->info.frame_stride) >> 8);
        k.rgba256 = (uint32_t)((MPEGHIP_SLOTS * rgba_stride_of(v)) >> 8);
        hipLaunchKernelGGL(replicate_desc_kernel, dim3((uint32_t)((work + 255) / 256)), dim3(256), 0, st, b->d_pics, n_pics,
                           descs ? b->d_mbs : nullptr, records ? b->d_xmbs : nullptr, n_mbs, k, replicas);
        HIP_TRY(hipGetLastError());
    }
    if (!sg) // pageable host memory: the copies above may still be reading it
        HIP_TRY(hipStreamSynchronize(st));
    b->notes.resize(n_pics);
    for (uint32_t p = 0; p < n_pics; p++) {
        b->notes[p].stream = pics[p].stream;
        b->notes[p].cur = pics[p].cur;
        b->notes[p].rgba = (pics[p].flags & MPEGHIP_PIC_RGBA) ? 1 : 0;
        b->notes[p].full = pics[p].mb_count == v->info.mb_w * v->info.mb_h ? 1 : 0; // (macroblocks of one submit do not overlap)
    }
    b->replicas = replicas;
    b->n_pics = (uint64_t)n_pics * replicas;
    b->n_mbs = (uint64_t)n_mbs * replicas;
    b->coef_bytes = (uint64_t)coef_bytes * replicas;
    b->alg_bytes *= replicas;
    return MPEGHIP_OK;
}

int mpeghip_video_submit(mpeghip_video *v, const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs,
                         uint32_t n_mbs, const void *coefs, size_t coef_bytes)
{
    if (!v)
        return fail(MPEGHIP_ERR_INVALID, "video is NULL");
    if (v->stage)
        return fail(MPEGHIP_ERR_INVALID, "submit while a stage is open (mpeghip_video_stage_commit ends it)");
    mpeghip_video::Staging *sg = &v->staging[v->next_staging];
    int rc = upload_into(v, &sg->batch, pics, n_pics, mbs, n_mbs, coefs, coef_bytes, 1, sg);
    if (rc != MPEGHIP_OK)
        return rc;
    v->next_staging ^= 1;
    rc = launch_batch(v, &sg->batch);
    if (rc != MPEGHIP_OK)
        return rc;
    HIP_TRY(hipEventRecord(sg->done, v->ctx->stream));
    sg->in_flight = true;
    return MPEGHIP_OK;
}

int mpeghip_video_stage_begin(mpeghip_video *v, uint32_t n_pics, const uint32_t *n_mbs, const size_t *coef_bytes,
                              mpeghip_stage **out)
{
    if (!v || !out || (n_pics && (!n_mbs || !coef_bytes)))
        return fail(MPEGHIP_ERR_INVALID, "stage_begin: NULL argument");
    if (v->stage)
        return fail(MPEGHIP_ERR_INVALID, "stage_begin: the previous stage is still open");
    if (recon_knob().mode != 6)
        return fail(MPEGHIP_ERR_INVALID, "stage_begin: staged submits feed the wave-chunk kernel only (MPEGHIP_RECON mode 6)");
    HIP_TRY(hipSetDevice(v->ctx->device));
    std::unique_ptr<mpeghip_stage> s(new mpeghip_stage);
    s->v = v;
    s->n_pics = n_pics;
    s->mb_first.resize(n_pics);
    s->mb_count.assign(n_mbs, n_mbs + n_pics);
    s->unit_first.resize(n_pics);
    s->units.resize(n_pics);
    s->alg.assign(n_pics, 0);
    s->done.assign(n_pics, 0);
    const size_t pb = sizeof(mpeghip_pic_desc) * (size_t)n_pics;
    uint64_t mbs = 0, units = 0, wire = 0; // wire: dwords, worst case (every unit dense) + headers
    for (uint32_t i = 0; i < n_pics; i++) {
        if (coef_bytes[i] % MPEGHIP_COEF_UNIT)
            return fail(MPEGHIP_ERR_INVALID, "stage_begin: picture %u: coef_bytes %zu is not a multiple of 128", i, coef_bytes[i]);
        s->mb_first[i] = (uint32_t)mbs;
        s->unit_first[i] = units;
        s->units[i] = coef_bytes[i] / MPEGHIP_COEF_UNIT;
        mbs += n_mbs[i];
        units += (s->units[i] + 7) & ~7ull; // a wave of wire_expand_kernel = 8 units of ONE picture
        wire += (s->units[i] * (1 + kWireUnitDwords) + 15) & ~15ull;
        if (mbs > 0xffffffffull || units > 0xffffffffull || wire > 0xffffffffull)
            return fail(MPEGHIP_ERR_INVALID, "batch too large for 32-bit descriptor indices");
    }
    s->n_mbs = (uint32_t)mbs;
    s->coef_units = units;
    s->wire_cap_dwords = wire;
    mpeghip_video::Staging *sg = &v->staging[v->next_staging];
    if (sg->in_flight) { // two submits ago: normally long finished
        HIP_TRY(hipEventSynchronize(sg->done));
        sg->in_flight = false;
    }
    const size_t xb = sizeof(uint32_t) * kXDwords * (size_t)mbs;
    s->tab_at = (pb + 63) & ~(size_t)63;
    s->x_at = (s->tab_at + sizeof(WireTab) * (size_t)n_pics + 63) & ~(size_t)63;
    s->wire0 = (s->x_at + xb + 63) & ~(size_t)63;
    const size_t need = s->wire0 + (size_t)wire * 4 + 64;
    if (need > sg->cap_h) {
        if (sg->h)
            (void)hipHostFree(sg->h);
        sg->h = nullptr;
        sg->cap_h = 0;
        const size_t cap = need + need / 2;
        HIP_TRY(hipHostMalloc((void **)&sg->h, cap, hipHostMallocDefault));
        sg->cap_h = cap;
    }
    if (!sg->done)
        HIP_TRY(hipEventCreateWithFlags(&sg->done, hipEventDisableTiming));
    s->sg = sg;
    v->stage = s.get();
    *out = s.release();
    return MPEGHIP_OK;
}

// Thread-safe for distinct i: touches only picture i's part of the staging buffer and of the stage's arrays.
int mpeghip_video_stage_put(mpeghip_stage *s, uint32_t i, const mpeghip_pic_desc *pic, const mpeghip_mb_desc *mbs,
                            const void *coefs)
{
    if (!s || !pic)
        return fail(MPEGHIP_ERR_INVALID, "stage_put: NULL argument");
    int rc = MPEGHIP_OK;
    do {
        if (i >= s->n_pics) {
            rc = fail(MPEGHIP_ERR_INVALID, "stage_put: picture %u of %u", i, s->n_pics);
            break;
        }
        const mpeghip_video *v = s->v;
        const uint32_t n = s->mb_count[i], first = s->mb_first[i];
        if ((n && !mbs) || (n && s->units[i] && !coefs)) {
            rc = fail(MPEGHIP_ERR_INVALID, "stage_put: picture %u: NULL array", i);
            break;
        }
        if ((rc = validate_pic(v->info, *pic, i)) != MPEGHIP_OK)
            break;
        uint8_t *h = s->sg->h;
        mpeghip_pic_desc pd = *pic;
        pd.mb_first = first;
        pd.mb_count = n;
        reinterpret_cast<mpeghip_pic_desc *>(h)[i] = pd;
        const XGeom geom = record_geometry(v);
        const XPic xp = expand_pic(geom, pd);
        uint32_t *xrec = reinterpret_cast<uint32_t *>(h + s->x_at);
        const uint32_t unit0 = (uint32_t)s->unit_first[i];
        uint64_t alg = 0;
        for (uint32_t k = 0; k < n && rc == MPEGHIP_OK; k++) {
            uint64_t units = 0;
            rc = validate_mb(v->info, pd, mbs[k], k, s->units[i], &units, &alg);
            if (rc != MPEGHIP_OK)
                break;
            uint32_t *x = xrec + (size_t)(first + k) * kXDwords;
            expand_mb(geom, xp, mbs[k], x);
            x[1] += unit0; // coef_off: relative to the picture's coefficients -> to the batch's
        }
        if (rc != MPEGHIP_OK)
            break;
        // horizontal runs of the chunks that lie inside this picture (commit looks at the straddling ones)
        for (uint32_t c = (first + kWcMbs - 1) / kWcMbs * kWcMbs; c + kWcMbs <= first + n; c += kWcMbs)
            mark_chunk_run(xrec, c);
        // the coefficient units, in wire form (video_wire_lane.h): headers, then the payload — packed in this
        // thread's scratch memory first, because the room a picture needs is only known afterwards
        const uint32_t units = (uint32_t)s->units[i];
        static thread_local std::vector<uint32_t> scratch;
        const size_t worst = (size_t)units * (1 + kWireUnitDwords);
        if (scratch.size() < worst)
            scratch.resize(worst + worst / 4 + 1024);
        uint32_t *hdr = scratch.data(), *payload = scratch.data() + units;
        uint32_t used = 0;
        const uint8_t *src = static_cast<const uint8_t *>(coefs);
        for (uint32_t u = 0; u < units; u++)
            hdr[u] = wire_pack_unit(src + (size_t)u * MPEGHIP_COEF_UNIT, payload, used);
        const uint32_t dwords = (units + used + 3) & ~3u; // regions stay 16-byte aligned
        const uint64_t at = s->wire_used.fetch_add(dwords);
        memcpy(h + s->wire0 + at * 4, scratch.data(), (size_t)(units + used) * 4);
        WireTab t;
        t.unit_first = unit0;
        t.units = units;
        t.region = (uint32_t)at;
        t.reserved = 0;
        reinterpret_cast<WireTab *>(h + s->tab_at)[i] = t;
        s->alg[i] = alg;
        s->done[i] = 1;
    } while (0);
    if (rc != MPEGHIP_OK) {
        std::lock_guard<std::mutex> l(s->error_lock);
        if (s->error.load() == MPEGHIP_OK) {
            s->error_text = mpeghip_last_error();
            s->error.store(rc);
        }
    }
    return rc;
}

int mpeghip_video_stage_commit(mpeghip_stage *sp)
{
    if (!sp)
        return fail(MPEGHIP_ERR_INVALID, "stage_commit: NULL stage");
    std::unique_ptr<mpeghip_stage> s(sp); // the stage ends here, whatever happens
    mpeghip_video *v = s->v;
    v->stage = nullptr;
    if (s->error.load() != MPEGHIP_OK)
        return fail(s->error.load(), "%s", s->error_text.c_str());
    for (uint32_t i = 0; i < s->n_pics; i++)
        if (!s->done[i])
            return fail(MPEGHIP_ERR_INVALID, "stage_commit: picture %u was never put", i);
    if (s->n_mbs == 0)
        return MPEGHIP_OK;
    HIP_TRY(hipSetDevice(v->ctx->device));
    mpeghip_video::Staging *sg = s->sg;
    mpeghip_batch *b = &sg->batch;
    const mpeghip_pic_desc *pics = reinterpret_cast<const mpeghip_pic_desc *>(sg->h);
    uint32_t *xrec = reinterpret_cast<uint32_t *>(sg->h + s->x_at);
    for (uint32_t i = 1; i < s->n_pics; i++) { // chunks that straddle two pictures
        const uint32_t c = s->mb_first[i] / kWcMbs * kWcMbs;
        if (c != s->mb_first[i] && c + kWcMbs <= s->n_mbs)
            mark_chunk_run(xrec, c);
    }
    b->any_rgba = wants_rgba(pics, s->n_pics);
    int rc;
    if (b->any_rgba && (rc = ensure_rgba(v)) != MPEGHIP_OK)
        return rc;
    // the device image of the staging buffer: pictures | WireTab | records | wire regions, sent in ONE copy
    // (a copy costs the better part of a millisecond of stream time whatever its size)
    const size_t cb = s->coef_units * MPEGHIP_COEF_UNIT;
    if (b->form != mpeghip_batch::StageBlob)
        batch_drop_descriptors(b); // (the batch was last used by a plain submit)
    b->form = mpeghip_batch::StageBlob;
    if ((rc = grow((void **)&b->d_blob, &b->cap_blob, s->wire0 + (size_t)s->wire_cap_dwords * 4 + 256)) != 0 ||
        (rc = grow((void **)&b->d_coefs, &b->cap_coefs, cb + 256)) != 0)
        return rc;
    b->d_pics = reinterpret_cast<mpeghip_pic_desc *>(b->d_blob);
    b->d_wtab = b->d_blob + s->tab_at;
    b->d_xmbs = reinterpret_cast<uint32_t *>(b->d_blob + s->x_at);
    b->d_wire = reinterpret_cast<uint32_t *>(b->d_blob + s->wire0);
    hipStream_t st = v->ctx->stream;
    {
        // in pieces: one copy of a gigabyte ran at a quarter of the rate of the same bytes in 32-128 MB pieces
        const size_t total = s->wire0 + (size_t)s->wire_used.load() * 4, piece = (size_t)64 << 20;
        for (size_t at = 0; at < total; at += piece)
            HIP_TRY(hipMemcpyAsync(b->d_blob + at, sg->h + at, total - at < piece ? total - at : piece, hipMemcpyHostToDevice, st));
    }
    uint32_t max_units = 0;
    for (uint32_t i = 0; i < s->n_pics; i++)
        max_units = s->units[i] > max_units ? (uint32_t)s->units[i] : max_units;
    if (max_units) {
        const uint32_t gx = ((max_units + 7) / 8 + 3) / 4;
        for (uint32_t p0 = 0; p0 < s->n_pics; p0 += 32768) {
            const uint32_t np = s->n_pics - p0 < 32768 ? s->n_pics - p0 : 32768;
            hipLaunchKernelGGL(wire_expand_kernel, dim3(gx, np), dim3(256), 0, st, b->d_wire,
                               static_cast<const WireTab *>(b->d_wtab), p0, b->d_coefs);
        }
        HIP_TRY(hipGetLastError());
    }
    b->notes.resize(s->n_pics);
    b->alg_bytes = 0;
    for (uint32_t p = 0; p < s->n_pics; p++) {
        b->notes[p].stream = pics[p].stream;
        b->notes[p].cur = pics[p].cur;
        b->notes[p].rgba = (pics[p].flags & MPEGHIP_PIC_RGBA) ? 1 : 0;
        b->notes[p].full = pics[p].mb_count == v->info.mb_w * v->info.mb_h ? 1 : 0;
        b->alg_bytes += s->alg[p];
    }
    b->replicas = 1;
    b->n_pics = s->n_pics;
    b->n_mbs = s->n_mbs;
    b->coef_bytes = cb;
    b->dense_partition = false;
    v->next_staging ^= 1;
    rc = launch_batch(v, b);
    if (rc != MPEGHIP_OK)
        return rc;
    HIP_TRY(hipEventRecord(sg->done, st));
    sg->in_flight = true;
    return MPEGHIP_OK;
}

int mpeghip_video_batch_upload_replicated(mpeghip_video *v, const mpeghip_pic_desc *pics, uint32_t n_pics,
                                          const mpeghip_mb_desc *mbs, uint32_t n_mbs, const void *coefs,
                                          size_t coef_bytes, uint32_t n_streams, mpeghip_batch **out)
{
    if (!v || !out)
        return fail(MPEGHIP_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_streams == 0 || n_streams > v->info.n_streams)
        return fail(MPEGHIP_ERR_INVALID, "n_streams %u out of range", n_streams);
    if (n_streams > 1)
        for (uint32_t p = 0; p < n_pics; p++)
            if (pics[p].stream != 0)
                return fail(MPEGHIP_ERR_INVALID, "replicated batches must describe stream 0");
    mpeghip_batch *b = new (std::nothrow) mpeghip_batch();
    if (!b)
        return fail(MPEGHIP_ERR_OOM, "host allocation failed");
    b->owner = v;
    int rc = upload_into(v, b, pics, n_pics, mbs, n_mbs, coefs, coef_bytes, n_streams);
    if (rc != MPEGHIP_OK) {
        batch_release(b);
        delete b;
        return rc;
    }
    *out = b;
    return MPEGHIP_OK;
}

int mpeghip_video_batch_upload(mpeghip_video *v, const mpeghip_pic_desc *pics, uint32_t n_pics,
                               const mpeghip_mb_desc *mbs, uint32_t n_mbs, const void *coefs, size_t coef_bytes,
                               mpeghip_batch **out)
{
    return mpeghip_video_batch_upload_replicated(v, pics, n_pics, mbs, n_mbs, coefs, coef_bytes, 1, out);
}

int mpeghip_video_batch_run(mpeghip_video *v, const mpeghip_batch *b)
{
    if (!v || !b || b->owner != v)
        return fail(MPEGHIP_ERR_INVALID, "batch does not belong to this video handle");
    HIP_TRY(hipSetDevice(v->ctx->device));
    return launch_batch(v, b);
}

void mpeghip_video_batch_free(mpeghip_batch *b)
{
    if (!b)
        return;
    if (b->owner) {
        (void)hipSetDevice(b->owner->ctx->device);
        (void)hipStreamSynchronize(b->owner->ctx->stream);
    }
    batch_release(b);
    delete b;
}

uint64_t mpeghip_video_batch_alg_bytes(const mpeghip_batch *b) { return b ? b->alg_bytes : 0; }
uint64_t mpeghip_video_batch_mbs(const mpeghip_batch *b) { return b ? b->n_mbs : 0; }

static uint8_t *slot_ptr(const mpeghip_video *v, uint32_t stream, uint32_t slot)
{
    return v->d_frames + ((uint64_t)stream * MPEGHIP_SLOTS + slot) * v->info.frame_stride;
}

void *mpeghip_video_slot_devptr(mpeghip_video *v, uint32_t stream, uint32_t slot)
{
    if (!v || stream >= v->info.n_streams || slot >= MPEGHIP_SLOTS)
        return nullptr;
    return slot_ptr(v, stream, slot);
}

void *mpeghip_video_rgba_devptr(mpeghip_video *v, uint32_t stream, uint32_t slot)
{
    if (!v || stream >= v->info.n_streams || slot >= MPEGHIP_SLOTS || ensure_rgba(v) != MPEGHIP_OK)
        return nullptr;
    return v->d_rgba + ((uint64_t)stream * MPEGHIP_SLOTS + slot) * rgba_stride_of(v);
}

int mpeghip_video_read_planes(mpeghip_video *v, uint32_t stream, uint32_t slot, uint8_t *y, uint8_t *cb, uint8_t *cr)
{
    if (!v || stream >= v->info.n_streams || slot >= MPEGHIP_SLOTS)
        return fail(MPEGHIP_ERR_INVALID, "bad stream/slot");
    HIP_TRY(hipSetDevice(v->ctx->device));
    // Y, Cb, Cr are one contiguous range of the slot: ONE copy into a pinned bounce buffer behind everything
    // queued on the stream, then plain memcpys (three synchronous copies into pageable memory cost three
    // round trips — most of a small picture's turnaround)
    const size_t bytes = v->info.luma_bytes + 2 * v->info.chroma_bytes;
    if (v->bounce_cap < bytes) {
        if (v->bounce)
            (void)hipHostFree(v->bounce);
        v->bounce = nullptr;
        v->bounce_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&v->bounce, bytes, hipHostMallocDefault));
        v->bounce_cap = bytes;
    }
    hipStream_t st = v->ctx->stream;
    HIP_TRY(hipMemcpyAsync(v->bounce, slot_ptr(v, stream, slot), bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (y)
        memcpy(y, v->bounce, v->info.luma_bytes);
    if (cb)
        memcpy(cb, v->bounce + v->info.luma_bytes, v->info.chroma_bytes);
    if (cr)
        memcpy(cr, v->bounce + v->info.luma_bytes + v->info.chroma_bytes, v->info.chroma_bytes);
    return MPEGHIP_OK;
}

int mpeghip_video_write_planes(mpeghip_video *v, uint32_t stream, uint32_t slot, const uint8_t *y, const uint8_t *cb,
                               const uint8_t *cr, const uint8_t *pad)
{
    if (!v || stream >= v->info.n_streams || slot >= MPEGHIP_SLOTS)
        return fail(MPEGHIP_ERR_INVALID, "bad stream/slot");
    v->rgba_sync[(size_t)stream * MPEGHIP_SLOTS + slot] = 0; // the slot's RGBA image is out of date now
    HIP_TRY(hipSetDevice(v->ctx->device));
    HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    uint8_t *p = slot_ptr(v, stream, slot);
    if (y)
        HIP_TRY(hipMemcpy(p, y, v->info.luma_bytes, hipMemcpyHostToDevice));
    if (cb)
        HIP_TRY(hipMemcpy(p + v->info.luma_bytes, cb, v->info.chroma_bytes, hipMemcpyHostToDevice));
    if (cr)
        HIP_TRY(hipMemcpy(p + v->info.luma_bytes + v->info.chroma_bytes, cr, v->info.chroma_bytes, hipMemcpyHostToDevice));
    if (pad)
        HIP_TRY(hipMemcpy(p + v->info.luma_bytes + 2 * v->info.chroma_bytes, pad, (size_t)v->info.luma_w * 16,
                          hipMemcpyHostToDevice));
    return MPEGHIP_OK;
}

int mpeghip_video_broadcast_slot(mpeghip_video *v, uint32_t src, uint32_t slot, uint32_t dst0, uint32_t n)
{
    if (!v || src >= v->info.n_streams || slot >= MPEGHIP_SLOTS || (uint64_t)dst0 + n > v->info.n_streams)
        return fail(MPEGHIP_ERR_INVALID, "bad stream/slot");
    HIP_TRY(hipSetDevice(v->ctx->device));
    for (uint32_t s = dst0; s < dst0 + n; s++)
        if (s != src)
            v->rgba_sync[(size_t)s * MPEGHIP_SLOTS + slot] = 0;
    for (uint32_t s = dst0; s < dst0 + n; s++) {
        if (s == src)
            continue;
        HIP_TRY(hipMemcpyAsync(slot_ptr(v, s, slot), slot_ptr(v, src, slot), v->info.frame_stride,
                               hipMemcpyDeviceToDevice, v->ctx->stream));
    }
    return MPEGHIP_OK;
}

int mpeghip_video_hash_slots(mpeghip_video *v, uint32_t slot, uint64_t *out)
{
    if (!v || !out || slot >= MPEGHIP_SLOTS)
        return fail(MPEGHIP_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(v->ctx->device));
    const uint64_t n_bytes = v->info.luma_bytes + 2 * v->info.chroma_bytes; // multiple of 128
    hipLaunchKernelGGL(hash_kernel, dim3((v->info.n_streams + 63) / 64), dim3(64), 0, v->ctx->stream, v->d_frames,
                       v->info.frame_stride, slot, n_bytes, v->info.n_streams, v->d_hash);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    HIP_TRY(hipMemcpy(out, v->d_hash, (size_t)v->info.n_streams * 8, hipMemcpyDeviceToHost));
    return MPEGHIP_OK;
}

int mpeghip_video_rgba_convert(mpeghip_video *v, uint32_t slot, uint32_t stream0, uint32_t n)
{
    if (!v || slot >= MPEGHIP_SLOTS || n == 0 || (uint64_t)stream0 + n > v->info.n_streams)
        return fail(MPEGHIP_ERR_INVALID, "bad stream/slot");
    HIP_TRY(hipSetDevice(v->ctx->device));
    if (!v->d_rgba) {
        // allocate without the initial conversion pass of ensure_rgba (we are it)
        const uint64_t total = rgba_stride_of(v) * MPEGHIP_SLOTS * v->info.n_streams;
        HIP_TRY(hipMalloc((void **)&v->d_rgba, total));
        for (uint32_t s = 0; s < MPEGHIP_SLOTS; s++)
            if (s != slot || stream0 != 0 || n != v->info.n_streams) {
                int rc = mpeghip_video_rgba_convert(v, s, 0, v->info.n_streams);
                if (rc != MPEGHIP_OK)
                    return rc;
            }
    }
    const mpeghip_video_info &in = v->info;
    const uint32_t quads = (in.width + 3) / 4;
    // grid.z is limited to 65535
    for (uint32_t s0 = 0; s0 < n; s0 += 32768) {
        const uint32_t ns = n - s0 < 32768 ? n - s0 : 32768;
        dim3 grid((quads + 63) / 64, (in.height + 7) / 8, ns);
        hipLaunchKernelGGL(rgba_kernel, grid, dim3(256), 0, v->ctx->stream, v->d_frames, in.frame_stride, v->d_rgba,
                           rgba_stride_of(v), in.luma_w, in.chroma_w, (uint32_t)in.luma_bytes, (uint32_t)in.chroma_bytes,
                           in.width, in.height, slot, stream0 + s0);
        HIP_TRY(hipGetLastError());
    }
    for (uint32_t st = stream0; st < stream0 + n; st++)
        v->rgba_sync[(size_t)st * MPEGHIP_SLOTS + slot] = 1;
    return MPEGHIP_OK;
}

int mpeghip_video_read_rgba(mpeghip_video *v, uint32_t stream, uint32_t slot, uint8_t *dst)
{
    if (!v || !dst || stream >= v->info.n_streams || slot >= MPEGHIP_SLOTS)
        return fail(MPEGHIP_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(v->ctx->device));
    int rc = ensure_rgba(v);
    if (rc != MPEGHIP_OK)
        return rc;
    HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    HIP_TRY(hipMemcpy(dst, v->d_rgba + ((uint64_t)stream * MPEGHIP_SLOTS + slot) * rgba_stride_of(v), v->info.rgba_bytes,
                      hipMemcpyDeviceToHost));
    return MPEGHIP_OK;
}

// -------------------------------------------------------------------- audio

int mpeghip_audio_open(mpeghip_ctx *c, uint32_t n_streams, int fma_mode, mpeghip_audio **out)
{
    if (!c || !out || n_streams == 0)
        return fail(MPEGHIP_ERR_INVALID, "bad argument");
    *out = nullptr;
    if (fma_mode != MPEGHIP_AUDIO_FMA_NONE && fma_mode != MPEGHIP_AUDIO_FMA_WINDOW)
        return fail(MPEGHIP_ERR_INVALID, "fma_mode %d", fma_mode);
    HIP_TRY(hipSetDevice(c->device));
    mpeghip_audio *a = new (std::nothrow) mpeghip_audio();
    if (!a)
        return fail(MPEGHIP_ERR_OOM, "host allocation failed");
    a->ctx = c;
    a->n_streams = n_streams;
    a->fma = fma_mode;
    float win[512];
    for (int i = 0; i < 512; i++)
        win[i] = (float)mpg_synth_window_x2[i] * 0.5f; // exact: entries are multiples of 0.5
    if (hipMalloc((void **)&a->d_ring, (size_t)n_streams * 2048 * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&a->d_vpos, (size_t)n_streams * sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void **)&a->d_ring_alt, (size_t)n_streams * 2048 * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&a->d_vpos_alt, (size_t)n_streams * sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void **)&a->d_window, sizeof(win)) != hipSuccess ||
        hipMemset(a->d_ring, 0, (size_t)n_streams * 2048 * sizeof(float)) != hipSuccess ||
        hipMemset(a->d_vpos, 0, (size_t)n_streams * sizeof(int32_t)) != hipSuccess ||
        hipMemcpy(a->d_window, win, sizeof(win), hipMemcpyHostToDevice) != hipSuccess) {
        mpeghip_audio_close(a);
        return fail(MPEGHIP_ERR_OOM, "audio state allocation failed");
    }
    if (getenv("MPEGHIP_DEBUG")) { // development aid: resident workgroups per CU of both kernels
        int na = 0, nv = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&na, audio_kernel<false, MPEGHIP_AUDIO_F32N>, kAudioThreads, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nv, recon_wc_kernel<4, false>, 256, 0);
        fprintf(stderr, "mpeghip: occupancy audio_kernel %d, recon_wc_kernel<4> %d workgroups per CU\n", na, nv);
    }
    *out = a;
    return MPEGHIP_OK;
}

void mpeghip_audio_close(mpeghip_audio *a)
{
    if (!a)
        return;
    (void)hipSetDevice(a->ctx->device);
    (void)hipStreamSynchronize(a->ctx->stream);
    void *ps[] = {a->d_ring, a->d_vpos, a->d_ring_alt, a->d_vpos_alt, a->d_window, a->d_samples, a->d_out};
    for (void *p : ps)
        if (p)
            (void)hipFree(p);
    if (a->d_active)
        (void)hipFree(a->d_active);
    delete a;
}

static size_t audio_elem_size(int format) { return format == MPEGHIP_AUDIO_S16 ? 2 : 4; }

int mpeghip_audio_device_buffers(mpeghip_audio *a, uint32_t n_frames, int format, int32_t **d_samples, void **d_out)
{
    if (!a || format < 0 || format > MPEGHIP_AUDIO_S16)
        return fail(MPEGHIP_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(a->ctx->device));
    HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    const size_t n = (size_t)a->n_streams * n_frames * MPEGHIP_AUDIO_FRAME_INTS;
    int rc;
    if ((rc = grow((void **)&a->d_samples, &a->cap_samples, n * sizeof(int32_t))) != 0 ||
        (rc = grow(&a->d_out, &a->cap_out, n * audio_elem_size(format))) != 0)
        return rc;
    if (d_samples)
        *d_samples = a->d_samples;
    if (d_out)
        *d_out = a->d_out;
    return MPEGHIP_OK;
}

int mpeghip_audio_upload(mpeghip_audio *a, int32_t *d_dst, const int32_t *src, size_t n_ints)
{
    if (!a || !d_dst || !src)
        return fail(MPEGHIP_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(a->ctx->device));
    HIP_TRY(hipMemcpy(d_dst, src, n_ints * sizeof(int32_t), hipMemcpyHostToDevice));
    return MPEGHIP_OK;
}

int mpeghip_audio_download(mpeghip_audio *a, void *dst, const void *d_src, size_t bytes)
{
    if (!a || !dst || !d_src)
        return fail(MPEGHIP_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(a->ctx->device));
    HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return MPEGHIP_OK;
}

static int audio_launch(mpeghip_audio *a, const int32_t *d_samples, uint32_t n_frames, int format, void *d_out, const uint8_t *d_active)
{
    if (!a || !d_samples || !d_out || format < 0 || format > MPEGHIP_AUDIO_S16)
        return fail(MPEGHIP_ERR_INVALID, "bad argument");
    if (n_frames == 0)
        return MPEGHIP_OK;
    HIP_TRY(hipSetDevice(a->ctx->device));
    AudioArgs args;
    args.samples = d_samples;
    args.out = d_out;
    args.ring = a->d_ring;
    args.vpos = a->d_vpos;
    args.ring_out = a->d_ring_alt;
    args.vpos_out = a->d_vpos_alt;
    args.window = a->d_window;
    args.n_streams = a->n_streams;
    args.n_frames = n_frames;
    args.format = format;
    args.fma = a->fma;
    args.active = d_active;
    // time slices per stream: one full residency of workgroups (4 per CU are resident in practice: 5 x 32 KB
    // of LDS do not fit next to the allocation granularity), at least 4 frames per slice
    uint32_t chunks = 1;
    if (const char *e = getenv("MPEGHIP_AUDIO_CHUNKS")) { // development knob
        chunks = (uint32_t)atoi(e);
    } else {
        int n_cu = 256;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, a->ctx->device);
        const uint32_t want = ((uint32_t)n_cu * 4 + a->n_streams - 1) / a->n_streams;
        chunks = want < 1 ? 1 : want;
        if (chunks > n_frames / 4)
            chunks = n_frames / 4;
    }
    if (chunks < 1)
        chunks = 1;
    if (chunks > n_frames)
        chunks = n_frames;
    args.n_chunks = chunks;
    const dim3 grid(a->n_streams * chunks), block(kAudioThreads);
#define LAUNCH_AUDIO(FMT)                                                                        \
    do {                                                                                         \
        if (args.fma)                                                                            \
            hipLaunchKernelGGL((audio_kernel<true, FMT>), grid, block, 0, a->ctx->stream, args);  \
        else                                                                                     \
            hipLaunchKernelGGL((audio_kernel<false, FMT>), grid, block, 0, a->ctx->stream, args); \
    } while (0)
    switch (format) {
    case MPEGHIP_AUDIO_F32N: LAUNCH_AUDIO(MPEGHIP_AUDIO_F32N); break;
    case MPEGHIP_AUDIO_F32NLR: LAUNCH_AUDIO(MPEGHIP_AUDIO_F32NLR); break;
    case MPEGHIP_AUDIO_S16: LAUNCH_AUDIO(MPEGHIP_AUDIO_S16); break;
    default: LAUNCH_AUDIO(MPEGHIP_AUDIO_F32); break;
    }
#undef LAUNCH_AUDIO
    HIP_TRY(hipGetLastError());
    { // the launch wrote the new state into the alternate buffers
        float *r = a->d_ring;
        a->d_ring = a->d_ring_alt;
        a->d_ring_alt = r;
        int32_t *v = a->d_vpos;
        a->d_vpos = a->d_vpos_alt;
        a->d_vpos_alt = v;
    }
    return MPEGHIP_OK;
}

int mpeghip_audio_synth_device(mpeghip_audio *a, const int32_t *d_samples, uint32_t n_frames, int format, void *d_out)
{
    return audio_launch(a, d_samples, n_frames, format, d_out, nullptr);
}

int mpeghip_audio_synth(mpeghip_audio *a, const int32_t *samples, uint32_t n_frames, int format, void *out)
{
    return mpeghip_audio_synth_masked(a, samples, n_frames, format, out, nullptr);
}

int mpeghip_audio_synth_masked(mpeghip_audio *a, const int32_t *samples, uint32_t n_frames, int format, void *out,
                               const uint8_t *active)
{
    if (!a || !samples || !out)
        return fail(MPEGHIP_ERR_INVALID, "NULL argument");
    if (n_frames == 0)
        return MPEGHIP_OK;
    int32_t *ds;
    void *dout;
    int rc = mpeghip_audio_device_buffers(a, n_frames, format, &ds, &dout);
    if (rc != MPEGHIP_OK)
        return rc;
    const size_t n = (size_t)a->n_streams * n_frames * MPEGHIP_AUDIO_FRAME_INTS;
    HIP_TRY(hipMemcpy(ds, samples, n * sizeof(int32_t), hipMemcpyHostToDevice));
    if (active) {
        if (!a->d_active)
            HIP_TRY(hipMalloc((void **)&a->d_active, a->n_streams));
        HIP_TRY(hipMemcpy(a->d_active, active, a->n_streams, hipMemcpyHostToDevice));
    }
    rc = audio_launch(a, ds, n_frames, format, dout, active ? a->d_active : nullptr);
    if (rc != MPEGHIP_OK)
        return rc;
    HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    HIP_TRY(hipMemcpy(out, dout, n * audio_elem_size(format), hipMemcpyDeviceToHost));
    return MPEGHIP_OK;
}

int mpeghip_audio_get_state(mpeghip_audio *a, uint32_t stream, float *v, int32_t *vpos)
{
    if (!a || stream >= a->n_streams)
        return fail(MPEGHIP_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(a->ctx->device));
    HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    if (v)
        HIP_TRY(hipMemcpy(v, a->d_ring + (size_t)stream * 2048, 2048 * sizeof(float), hipMemcpyDeviceToHost));
    if (vpos)
        HIP_TRY(hipMemcpy(vpos, a->d_vpos + stream, sizeof(int32_t), hipMemcpyDeviceToHost));
    return MPEGHIP_OK;
}

int mpeghip_audio_set_state(mpeghip_audio *a, uint32_t stream, const float *v, int32_t vpos)
{
    if (!a || stream >= a->n_streams || vpos < 0 || vpos > 1023 || (vpos & 63))
        return fail(MPEGHIP_ERR_INVALID, "bad argument");
    if (v) {
        // Audio.v only ever holds idct36 outputs (or zeros): each 64-entry slot is the signed mirror
        // of 32 DCT outputs (audio.go:708-771).  The kernel keeps just those 32, so insist on it.
        for (int ch = 0; ch < 2; ch++)
            for (int slot = 0; slot < 16; slot++) {
                const float *d = v + ch * 1024 + slot * 64;
                bool ok = d[16] == 0.0f && d[0] == -d[32];
                for (int k = 1; k <= 15 && ok; k++)
                    ok = d[48 + k] == d[48 - k] && d[16 - k] == -d[16 + k];
                if (!ok)
                    return fail(MPEGHIP_ERR_INVALID, "v is not a synthesis state (channel %d slot %d breaks the idct36 mirror)", ch, slot);
            }
    }
    HIP_TRY(hipSetDevice(a->ctx->device));
    HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    if (v)
        HIP_TRY(hipMemcpy(a->d_ring + (size_t)stream * 2048, v, 2048 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(a->d_vpos + stream, &vpos, sizeof(int32_t), hipMemcpyHostToDevice));
    return MPEGHIP_OK;
}

#ifdef MPG_PHASE_TIMING
int mpeghip_debug_read_dump(mpeghip_video *v, void *dst, size_t bytes)
{
    HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    HIP_TRY(hipMemcpy(dst, v->d_dump, bytes, hipMemcpyDeviceToHost));
    return MPEGHIP_OK;
}
#endif

} // extern "C"
