// emu.cpp — TEST INFRASTRUCTURE ONLY: a CPU "lane emulator" for the kernels of
// libmpeghip.  It compiles the very same lane functions the GPU kernels are made
// of (mpeg_amd/csrc/*_lane.h) with g++ and runs the 64 lanes of a wavefront /
// the 256 threads of an audio workgroup in plain loops, phase by phase, with LDS
// as an ordinary array.  Purpose: check lane mapping, addressing and the integer
// range invariants (MPG_EMU_CHECKS) against the oracle on a machine without a
// GPU, under ASan/UBSan if wanted.
//
// It is NOT a fallback: it is built into tests/kernel_emu/libkernel_emu.so, which
// only tests load; the product library has no CPU path and fails without a GPU.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "audio_lane.h"
#include "video_lane.h"
#include "video_split_lane.h"
#include "video_compact_lane.h"
#include "video_wire_lane.h"

using namespace mpg;

extern "C" {

// Reconstruct n_mbs macroblocks exactly as recon_kernel<W> does, one "wave" at a time.
int emu_video_run(uint8_t *frames, uint64_t frame_stride, uint32_t luma_w, uint32_t luma_h,
                  uint32_t width, uint32_t height,
                  const mpeghip_pic_desc *pics, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                  const uint8_t *coefs, const uint8_t *qtable, uint8_t *dump,
                  uint8_t *rgba, uint64_t rgba_stride, int static_pipeline)
{
    VideoArgs a;
    a.frames = frames;
    a.frame_stride = frame_stride;
    a.luma_w = luma_w;
    a.luma_h = luma_h;
    a.chroma_w = luma_w / 2;
    a.chroma_h = luma_h / 2;
    a.luma_bytes = luma_w * luma_h;
    a.chroma_bytes = a.luma_bytes / 4;
    a.pics = pics;
    a.mbs = mbs;
    a.coefs = coefs;
    a.qmat = qtable;
    a.dump = dump;
    a.n_mbs = n_mbs;
    a.width = width;
    a.height = height;
    a.rgba = rgba;
    a.rgba_stride = rgba_stride;

    alignas(16) int32_t tile[kTileDwords];
    alignas(16) uint8_t stage[kRgbaBytes];
    // walk the grid the way the kernel does (8 waves per block, XCD remap) so the
    // chunk mapping is exercised too; the result must not depend on the order
    const uint32_t WAVES = 8;
    const uint32_t blocks = (n_mbs + WAVES - 1) / WAVES;
    for (uint32_t blk = 0; blk < blocks; blk++) {
        const uint32_t chunk = xcd_chunk(blk, blocks);
        for (uint32_t wave = 0; wave < WAVES; wave++) {
            const uint32_t mb_index = chunk * WAVES + wave;
            if (mb_index >= n_mbs)
                continue;
            const MbU u = load_mb(a, mb_index);
            MbLane st[64];
            memset(tile, 0xCD, sizeof(tile)); // poison: reads of unwritten LDS must not matter
            uint64_t out[64];
            bool wrote[64];
            if (static_pipeline) { // the lane functions of recon_kernel<W, 3>
                MbLoads ld[64];
                for (int lane = 0; lane < 64; lane++)
                    mb_issue_loads_static(a, u, lane, ld[lane]);
                for (int lane = 0; lane < 64; lane++)
                    mb_phase_a_compute_static(a, u, lane, ld[lane], st[lane], tile);
                for (int lane = 0; lane < 64; lane++)
                    out[lane] = mb_phase_b_t<true>(a, u, lane, st[lane], tile, wrote[lane], dump + lane * 8);
            } else {
                for (int lane = 0; lane < 64; lane++)
                    mb_phase_a(a, u, lane, st[lane], tile);
                for (int lane = 0; lane < 64; lane++)
                    out[lane] = mb_phase_b(a, u, lane, st[lane], tile, wrote[lane]);
            }
            if (u.rgba) {
                for (int lane = 0; lane < 64; lane++)
                    mb_phase_c_stage(a, u, lane, out[lane], wrote[lane], stage);
                for (int lane = 0; lane < 64; lane++)
                    mb_phase_c_convert(a, u, lane, stage);
            }
        }
    }
    return 0;
}

// The split path: pred_kernel<W> then resid_kernel<W> (then the RGBA pass for flagged pictures).
int emu_video_run_split(uint8_t *frames, uint64_t frame_stride, uint32_t luma_w, uint32_t luma_h,
                        uint32_t width, uint32_t height,
                        const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                        const uint8_t *coefs, uint32_t n_units, const uint8_t *qtable,
                        uint8_t *rgba, uint64_t rgba_stride)
{
    SplitArgs s;
    VideoArgs &a = s.v;
    a.frames = frames;
    a.frame_stride = frame_stride;
    a.luma_w = luma_w;
    a.luma_h = luma_h;
    a.chroma_w = luma_w / 2;
    a.chroma_h = luma_h / 2;
    a.luma_bytes = luma_w * luma_h;
    a.chroma_bytes = a.luma_bytes / 4;
    a.pics = pics;
    a.mbs = mbs;
    a.coefs = coefs;
    a.qmat = qtable;
    a.dump = nullptr;
    a.n_mbs = n_mbs;
    a.width = width;
    a.height = height;
    a.rgba = rgba;
    a.rgba_stride = rgba_stride;
    std::vector<BlockEntry> entries(n_units ? n_units : 1);
    memset(entries.data(), 0xff, entries.size() * sizeof(BlockEntry));
    s.entries = entries.data();
    s.n_units = n_units;
    const uint32_t WAVES = 8;
    // K1
    {
        const uint32_t blocks = (n_mbs + 2 * WAVES - 1) / (2 * WAVES);
        for (uint32_t blk = 0; blk < blocks; blk++) {
            const uint32_t chunk = xcd_chunk(blk, blocks);
            for (uint32_t wave = 0; wave < WAVES; wave++)
                for (int lane = 0; lane < 64; lane++) {
                    const uint32_t mb_index = (chunk * WAVES + wave) * 2 + (uint32_t)(lane >> 5);
                    if (mb_index < n_mbs)
                        pred_lane(s, load_pred_mb(s.v, mb_index), lane & 31);
                }
        }
    }
    // K2
    {
        const uint32_t blocks = (n_units + 8 * WAVES - 1) / (8 * WAVES);
        alignas(16) int32_t tile[kResidTileDwords];
        for (uint32_t blk = 0; blk < blocks; blk++) {
            const uint32_t chunk = xcd_chunk(blk, blocks);
            for (uint32_t wave = 0; wave < WAVES; wave++) {
                ResidLane st[64];
                memset(tile, 0xCD, sizeof(tile));
                for (int lane = 0; lane < 64; lane++) {
                    const int g = lane >> 3, j = lane & 7;
                    resid_phase_a(s, (chunk * WAVES + wave) * 8 + (uint32_t)g, j, tile + g * kTileStride, st[lane]);
                }
                for (int lane = 0; lane < 64; lane++) {
                    const int g = lane >> 3, j = lane & 7;
                    resid_phase_b(s, j, tile + g * kTileStride, st[lane]);
                }
            }
        }
    }
    // RGBA pass
    for (uint32_t p = 0; p < n_pics; p++) {
        if (!(pics[p].flags & MPEGHIP_PIC_RGBA))
            continue;
        const uint64_t fs = (uint64_t)pics[p].stream * MPEGHIP_SLOTS + pics[p].cur;
        const uint32_t quads = (width + 3) / 4;
        for (uint32_t y = 0; y < ((height + 7) / 8) * 4; y++) // row pairs
            for (uint32_t x4 = 0; x4 < ((quads + 63) / 64) * 64; x4++)
                rgba_convert_quad(frames + fs * frame_stride, a.luma_w, a.chroma_w, a.luma_bytes, a.chroma_bytes, width, height,
                                  x4, y, rgba + fs * rgba_stride);
    }
    return 0;
}

// recon_compact_kernel: one workgroup (8 waves) per chunk of 8 macroblocks (+ the RGBA pass).
int emu_video_run_compact(uint8_t *frames, uint64_t frame_stride, uint32_t luma_w, uint32_t luma_h,
                          uint32_t width, uint32_t height,
                          const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                          const uint8_t *coefs, const uint8_t *qtable, uint8_t *rgba, uint64_t rgba_stride)
{
    VideoArgs a;
    a.frames = frames;
    a.frame_stride = frame_stride;
    a.luma_w = luma_w;
    a.luma_h = luma_h;
    a.chroma_w = luma_w / 2;
    a.chroma_h = luma_h / 2;
    a.luma_bytes = luma_w * luma_h;
    a.chroma_bytes = a.luma_bytes / 4;
    a.pics = pics;
    a.mbs = mbs;
    a.coefs = coefs;
    a.qmat = qtable;
    a.dump = nullptr;
    a.n_mbs = n_mbs;
    a.width = width;
    a.height = height;
    a.rgba = rgba;
    a.rgba_stride = rgba_stride;
    alignas(16) static uint8_t lds[kCompactLdsBytes];
    const uint32_t blocks = (n_mbs + kChunkMbs - 1) / kChunkMbs;
    for (uint32_t blk = 0; blk < blocks; blk++) {
        const uint32_t chunk = xcd_chunk(blk, blocks);
        const ChunkInfo ci = load_chunk(a, chunk);
        memset(lds, 0xCD, sizeof(lds));
        MbU u[kChunkMbs];
        static MbLoads ld[kChunkMbs][64];
        for (uint32_t w = 0; w < (uint32_t)kChunkMbs; w++) {
            if (w >= ci.n)
                continue;
            u[w] = load_mb(a, chunk * kChunkMbs + w);
            for (int lane = 0; lane < 64; lane++)
                compact_phase1(a, u[w], lane, ld[w][lane]);
        }
        for (uint32_t w = 0; w < (uint32_t)kChunkMbs; w++) {
            if (8 * w >= ci.base[kChunkMbs])
                continue;
            bool active[64];
            for (int lane = 0; lane < 64; lane++) {
                const int g = lane >> 3, j = lane & 7;
                int32_t *tile_g = reinterpret_cast<int32_t *>(lds + kResidStoreBytes) + (w * 8 + (uint32_t)g) * kTileStride;
                compact_phase2(a, ci, 8 * w + (uint32_t)g, j, tile_g, active[lane]);
            }
            for (int lane = 0; lane < 64; lane++) {
                const int g = lane >> 3, j = lane & 7;
                const int32_t *tile_g = reinterpret_cast<const int32_t *>(lds + kResidStoreBytes) + (w * 8 + (uint32_t)g) * kTileStride;
                compact_phase2_rows(8 * w + (uint32_t)g, j, tile_g, active[lane], lds);
            }
        }
        for (uint32_t w = 0; w < (uint32_t)kChunkMbs; w++) {
            if (w >= ci.n)
                continue;
            for (int lane = 0; lane < 64; lane++)
                compact_phase3(a, u[w], ci, w, lane, ld[w][lane], lds);
        }
    }
    for (uint32_t p = 0; p < n_pics; p++) {
        if (!(pics[p].flags & MPEGHIP_PIC_RGBA))
            continue;
        const uint64_t fs = (uint64_t)pics[p].stream * MPEGHIP_SLOTS + pics[p].cur;
        const uint32_t quads = (width + 3) / 4;
        for (uint32_t y = 0; y < ((height + 7) / 8) * 4; y++) // row pairs
            for (uint32_t x4 = 0; x4 < ((quads + 63) / 64) * 64; x4++)
                rgba_convert_quad(frames + fs * frame_stride, a.luma_w, a.chroma_w, a.luma_bytes, a.chroma_bytes, width, height,
                                  x4, y, rgba + fs * rgba_stride);
    }
    return 0;
}

// recon_wc_kernel: one wave per chunk of 4 macroblocks (+ the RGBA pass).
int emu_video_run_wc(uint8_t *frames, uint64_t frame_stride, uint32_t luma_w, uint32_t luma_h,
                     uint32_t width, uint32_t height,
                     const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                     const uint8_t *coefs, const uint8_t *qtable, uint8_t *rgba, uint64_t rgba_stride)
{
    VideoArgs a;
    a.frames = frames;
    a.frame_stride = frame_stride;
    a.luma_w = luma_w;
    a.luma_h = luma_h;
    a.chroma_w = luma_w / 2;
    a.chroma_h = luma_h / 2;
    a.luma_bytes = luma_w * luma_h;
    a.chroma_bytes = a.luma_bytes / 4;
    // the library's expanded records (built by the host half of the product during validation)
    XGeom geom;
    geom.luma_w = a.luma_w;
    geom.chroma_w = a.chroma_w;
    geom.frame_stride = frame_stride;
    geom.rgba_stride = rgba_stride;
    bool any_rgba = false; // the product picks the kernel instance by this (mpeghip.hip: launch_batch)
    for (uint32_t p = 0; p < n_pics; p++)
        any_rgba = any_rgba || (pics[p].flags & MPEGHIP_PIC_RGBA);
    if (frame_stride % 256 || rgba_stride % 256)
        abort(); // the records name frames in units of 256 bytes
    std::vector<uint32_t> xrec((size_t)n_mbs * kXDwords + 16);
    for (uint32_t i = 0; i < n_mbs; i++)
        expand_mb(geom, expand_pic(geom, pics[mbs[i].pic]), mbs[i], xrec.data() + (size_t)i * kXDwords);
    mark_chunk_runs(xrec.data(), n_mbs);
    a.pics = pics;
    a.mbs = mbs;
    a.xmbs = xrec.data();
    a.coefs = coefs;
    a.qmat = qtable;
    a.dump = nullptr;
    a.n_mbs = n_mbs;
    a.width = width;
    a.height = height;
    a.rgba = rgba;
    a.rgba_stride = rgba_stride;
    alignas(16) static uint8_t lds[kWcLdsBytes];
    const uint32_t n_chunks = (n_mbs + kWcMbs - 1) / kWcMbs;
    for (uint32_t chunk = 0; chunk < n_chunks; chunk++) {
        memset(lds, 0xCD, sizeof(lds));
        uint8_t *resid = lds;
        int32_t *tile = reinterpret_cast<int32_t *>(lds + kWcResidBytes);
        uint32_t n_live;
        WcRaw raw;
        if (any_rgba)
            wc_load_raw<true>(a, chunk, n_live, raw);
        else
            wc_load_raw<false>(a, chunk, n_live, raw);
        const WcInfo ci = wc_info_from_raw(n_live, raw);
        MbU u[kWcMbs];
        static MbLoads ld[kWcMbs][64];
        for (int m = 0; m < kWcMbs; m++) {
            u[m] = any_rgba ? wc_mb_from_raw<true>(a, raw.d[m]) : wc_mb_from_raw<false>(a, raw.d[m]);
            for (int lane = 0; lane < 64; lane++)
                wc_issue_pred(a, u[m], lane, ld[m][lane]);
        }
        for (uint32_t s0 = 0; s0 < ci.base[kWcMbs]; s0 += 8) {
            bool active[64];
            for (int lane = 0; lane < 64; lane++)
                compact_phase2(a, ci, s0 + (uint32_t)(lane >> 3), lane & 7, tile + (lane >> 3) * kWcTileStride, active[lane]);
            for (int lane = 0; lane < 64; lane++)
                compact_phase2_rows(s0 + (uint32_t)(lane >> 3), lane & 7, tile + (lane >> 3) * kWcTileStride, active[lane], resid);
        }
        const bool coalesce = wc_can_coalesce(ci, u);
        bool rgba_any = false;
        for (int m = 0; m < kWcMbs; m++)
            rgba_any = rgba_any || ((uint32_t)m < ci.n && u[m].rgba != nullptr);
        uint8_t *out_tile = (coalesce || rgba_any) ? reinterpret_cast<uint8_t *>(tile) : nullptr;
        if (out_tile)
            memset(tile, 0xEE, kWcTileBytes);
        for (int m = 0; m < kWcMbs; m++) {
            if ((uint32_t)m >= ci.n)
                continue;
            for (int lane = 0; lane < 64; lane++) {
                u8x16 below = ld[m][lane].r1;
                const int bl = wc_below_lane(lane);
                if (wc_needs_below(u[m]) && bl >= 0) // the kernel's __shfl from the owning lane
                    for (int k = 0; k < 3; k++)
                        below.v[k] = ld[m][bl].r0.v[k];
                if (coalesce)
                    wc_phase3<kWcMbs, true>(a, u[m], ci, (uint32_t)m, lane, ld[m][lane], below, resid, out_tile, false);
                else
                    wc_phase3<kWcMbs, false>(a, u[m], ci, (uint32_t)m, lane, ld[m][lane], below, resid, out_tile, true);
            }
        }
        if (coalesce)
            for (int lane = 0; lane < 64; lane++)
                wc_store_tile(a, u[0], lane, out_tile);
        if (rgba_any) // Frame.RGBA of the written macroblocks, fused (the host adds a whole-frame pass only
            for (int m = 0; m < kWcMbs; m++) // for partial pictures over a slot whose image is out of date)
                if ((uint32_t)m < ci.n && u[m].rgba != nullptr)
                    for (int lane = 0; lane < 64; lane++)
                        wc_rgba_mb(a, u[m], (uint32_t)m, lane, out_tile);
    }
    return 0;
}

// rgba_pixel (the arrangement the device uses) against ycbcr_to_rgba (the reference's form) for ALL
// 2^24 (y, cb, cr): returns the number of disagreements
uint32_t emu_rgba_forms_disagree(void)
{
    uint32_t bad = 0;
    for (uint32_t cb = 0; cb < 256; cb++)
        for (uint32_t cr = 0; cr < 256; cr++) {
            const ChromaTerms c = chroma_terms(cb, cr);
            for (uint32_t y = 0; y < 256; y += 4) {
                const uint32_t yw = y | ((y + 1) << 8) | ((y + 2) << 16) | ((y + 3) << 24);
                uint32_t px[4];
                rgba_row4(yw, c, c, px);
                for (uint32_t k = 0; k < 4; k++)
                    bad += px[k] != ycbcr_to_rgba(y + k, cb, cr);
            }
        }
    return bad;
}

void emu_rgba_convert(const uint8_t *frame, uint32_t luma_w, uint32_t luma_h, uint32_t width, uint32_t height,
                      uint8_t *rgba)
{
    const uint32_t quads = (width + 3) / 4;
    for (uint32_t y = 0; y < ((height + 7) / 8) * 4; y++) // row pairs
        for (uint32_t x4 = 0; x4 < ((quads + 63) / 64) * 64; x4++)
            rgba_convert_quad(frame, luma_w, luma_w / 2, luma_w * luma_h, luma_w * luma_h / 4, width, height, x4, y, rgba);
}

} // extern "C"

// audio_kernel: n_chunks workgroups (time slices) per stream; ring / vpos are updated in place for the
// caller (the kernel writes them to the alternate buffers, emulated with a copy).  Between two
// barriers the kernel's waves run different phases concurrently; the emulation runs them in the
// order that would expose a hazard (the DCTs of step s+1 BEFORE the windows of step s).
template <bool kFma, int kFormat> static void emu_audio_blocks(const AudioArgs &a)
{
    std::vector<float> lds(kAudioLdsFloats);
    struct Regs { float dreg[16]; };
    std::vector<Regs> regs(kAudioThreads);
    for (uint32_t blk = 0; blk < a.n_streams * a.n_chunks; blk++) {
        const uint32_t stream = blk / a.n_chunks, chunk = blk % a.n_chunks;
        uint32_t f0, f1;
        audio_chunk_range(a, chunk, f0, f1);
        if (f0 >= f1)
            continue;
        if (a.active && a.active[stream] == 0) {
            if (f1 == a.n_frames)
                for (int tid = 0; tid < kAudioThreads; tid++)
                    audio_carry_state(a, stream, tid);
            continue;
        }
        for (auto &x : lds)
            x = 1e30f; // poison
        const int32_t vpos0 = a.vpos[stream];
        const uint32_t tg0 = f0 * 36, tg1 = f1 * 36, n_steps = (tg1 - tg0 + kStep - 1) / kStep;
        for (int tid = 0; tid < kAudioThreads; tid++) {
            audio_load_window(a, tid, regs[tid].dreg);
            audio_phase_fetch(a, stream, tg0, tg1, 0, tid, lds.data());
            if (f0 == 0)
                audio_load_state(a, stream, vpos0, tid, lds.data());
            else
                audio_phase_warmup(a, stream, f0, tid, lds.data());
        }
        for (int tid = 0; tid < kAudioThreads; tid++)
            audio_phase_dct(a, stream, tg0, tg1, 0, tid, lds.data());
        for (uint32_t si = 0; si < n_steps; si++) {
            for (int tid = 0; tid < kAudioThreads; tid++)
                audio_phase_dct(a, stream, tg0, tg1, si + 1, tid, lds.data());
            for (int tid = 0; tid < kAudioThreads; tid++)
                audio_phase_window<kFma, kFormat>(a, stream, vpos0, tg0, tg1, si, tid, regs[tid].dreg, lds.data());
        }
        if (f1 == a.n_frames) {
            for (int tid = 0; tid < kAudioThreads; tid++)
                audio_store_state(a, stream, vpos0, tid, lds.data());
            audio_store_vpos(a, stream, vpos0);
        }
    }
}

template <bool kFma> static void emu_audio_format(const AudioArgs &a)
{
    switch (a.format) {
    case MPEGHIP_AUDIO_F32N: return emu_audio_blocks<kFma, MPEGHIP_AUDIO_F32N>(a);
    case MPEGHIP_AUDIO_F32NLR: return emu_audio_blocks<kFma, MPEGHIP_AUDIO_F32NLR>(a);
    case MPEGHIP_AUDIO_S16: return emu_audio_blocks<kFma, MPEGHIP_AUDIO_S16>(a);
    default: return emu_audio_blocks<kFma, MPEGHIP_AUDIO_F32>(a);
    }
}

extern "C" {
int emu_audio_run_masked(const int32_t *samples, void *out, float *ring, int32_t *vpos, const float *window,
                         uint32_t n_streams, uint32_t n_frames, int32_t format, int32_t fma, uint32_t n_chunks, const uint8_t *active);
int emu_audio_run(const int32_t *samples, void *out, float *ring, int32_t *vpos, const float *window,
                  uint32_t n_streams, uint32_t n_frames, int32_t format, int32_t fma, uint32_t n_chunks)
{
    return emu_audio_run_masked(samples, out, ring, vpos, window, n_streams, n_frames, format, fma, n_chunks, nullptr);
}
int emu_audio_run_masked(const int32_t *samples, void *out, float *ring, int32_t *vpos, const float *window,
                         uint32_t n_streams, uint32_t n_frames, int32_t format, int32_t fma, uint32_t n_chunks, const uint8_t *active)
{
    if (n_frames == 0)
        return 0;
    std::vector<float> ring_out((size_t)n_streams * 2048);
    std::vector<int32_t> vpos_out(n_streams);
    AudioArgs a;
    a.samples = samples;
    a.out = out;
    a.ring = ring;
    a.vpos = vpos;
    a.ring_out = ring_out.data();
    a.vpos_out = vpos_out.data();
    a.window = window;
    a.n_streams = n_streams;
    a.n_frames = n_frames;
    a.format = format;
    a.fma = fma;
    a.active = active;
    a.n_chunks = n_chunks < 1 ? 1 : (n_chunks > n_frames ? n_frames : n_chunks);
    if (fma)
        emu_audio_format<true>(a);
    else
        emu_audio_format<false>(a);
    memcpy(ring, ring_out.data(), ring_out.size() * sizeof(float));
    memcpy(vpos, vpos_out.data(), vpos_out.size() * sizeof(int32_t));
    return 0;
}

// scalar helpers exposed for unit tests
uint32_t emu_avg4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return avg4_u8x4(a, b, c, d); }
uint32_t emu_avg2(uint32_t a, uint32_t b) { return avg_ceil_u8x4(a, b); }
uint32_t emu_xcd_chunk(uint32_t b, uint32_t n) { return xcd_chunk(b, n); }
uint32_t emu_ycbcr(uint32_t y, uint32_t cb, uint32_t cr) { return ycbcr_to_rgba(y, cb, cr); }

// video_wire_lane.h: pack n dense units the way mpeghip_video_stage_put does, then rebuild them the way
// wire_expand_kernel's lanes do.  Returns the dwords on the wire (headers + payload); out = n * 128 bytes.
uint32_t emu_wire_roundtrip(const uint8_t *units, uint32_t n, uint8_t *out)
{
    std::vector<uint32_t> region((size_t)n * (1 + kWireUnitDwords) + 16);
    uint32_t *hdr = region.data(), *payload = region.data() + n;
    uint32_t used = 0;
    for (uint32_t u = 0; u < n; u++)
        hdr[u] = wire_pack_unit(units + (size_t)u * 128, payload, used);
    alignas(16) uint8_t tile[1024];
    for (uint32_t group = 0; group * 8 < n; group++) {
        WireLane w[64];
        memset(tile, 0xCD, sizeof(tile));
        for (int lane = 0; lane < 64; lane++) {
            const uint32_t unit = group * 8 + (uint32_t)(lane >> 3);
            w[lane].live = unit < n;
            w[lane].header = w[lane].live ? hdr[unit] : 0;
            w[lane].payload = payload;
            wire_phase_zero(tile, lane);
        }
        for (int lane = 0; lane < 64; lane++)
            wire_phase_scatter(w[lane], tile, lane);
        for (int lane = 0; lane < 64; lane++)
            wire_phase_store(w[lane], tile, lane, out + (size_t)(group * 8 + (uint32_t)(lane >> 3)) * 128);
    }
    return n + used;
}
}
