// video_compact_lane.h — fused reconstruction with a DENSE residual stage.
//
// One workgroup = 8 waves = one chunk of 8 consecutive macroblocks.
//   phase 1  wave w, lane (b, j): issue the prediction loads of ITS macroblock (row j of
//            block b) — they stay in flight during phase 2.
//   phase 2  the chunk's coded blocks (at most 48, typically ~15) are numbered 0..T-1 in
//            (macroblock, block) order; wave w takes slots 8w..8w+7, lane (g, j) = column j
//            of slot 8w+g: dequantise, column pass, wave-private LDS transpose, row pass,
//            saturate to int16 and park the residual ROW in the chunk's LDS residual store.
//            Waves with 8w >= T skip straight to the barrier.  Every lane of a working wave
//            has a coded block: the IDCT instruction stream is paid once per 8 coded blocks
//            instead of once per macroblock (measured: 61 % of macroblock-waves ran it with
//            at most half their lanes coded).
//   barrier
//   phase 3  wave w, lane (b, j): prediction average, + residual row from LDS (if coded),
//            clamp, one 8-byte store.  Destination bytes are written exactly once and never
//            read: no traffic beyond the fused kernel's.
//
// Same arithmetic, same results as video_lane.h (bit-exact; the int16 saturation of the
// residual is exact because |residual| >= 32767 saturates the pixel either way).
#pragma once

#include "video_lane.h"

namespace mpg {

constexpr int kChunkMbs = 8;
constexpr int kMaxChunkBlocks = 6 * kChunkMbs;                    // 48
constexpr int kResidStoreBytes = kMaxChunkBlocks * 128;            // int16 8x8 per slot
constexpr int kCompactTileBytes = 6 * 8 * kTileStride * 4;         // 6 working waves x 8 blocks x 72 dwords
constexpr int kCompactLdsBytes = kResidStoreBytes + kCompactTileBytes;

// Wave-uniform summary of a chunk of N consecutive descriptors (SGPRs on the device).
template <int N>
struct ChunkInfoT {
    uint32_t n;               // macroblocks in this chunk (tail chunk < N)
    uint32_t cbp[N];          // 0 for macroblocks beyond n
    uint32_t flags[N];        // MPEGHIP_MB_* | quantiser_scale << 8
    uint32_t coef_off[N];
    uint32_t qtab[N];         // byte offset of the macroblock's {matrix, premultiplier} table
    uint32_t base[N + 1];     // slot number of each macroblock's first coded block; base[N] = total
};
using ChunkInfo = ChunkInfoT<kChunkMbs>;

template <int N>
MPG_HD ChunkInfoT<N> load_chunk_t(const VideoArgs &a, uint32_t chunk)
{
    const MPG_CONST_AS mpeghip_mb_desc *mbs = (const MPG_CONST_AS mpeghip_mb_desc *)(uintptr_t)a.mbs;
    const MPG_CONST_AS mpeghip_pic_desc *pics = (const MPG_CONST_AS mpeghip_pic_desc *)(uintptr_t)a.pics;
    ChunkInfoT<N> ci;
    const uint32_t first = chunk * N;
    ci.n = a.n_mbs - first < (uint32_t)N ? a.n_mbs - first : (uint32_t)N;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t idx = (uint32_t)k < ci.n ? first + (uint32_t)k : first; // clamp: keeps the loads unconditional
        const MPG_CONST_AS mpeghip_mb_desc &d = mbs[idx];
        // flags | cbp << 8 | qscale << 16 as ONE scalar dword (byte-sized member reads would become
        // per-descriptor vector byte loads)
        const uint32_t w3 = *(const MPG_CONST_AS uint32_t *)((const MPG_CONST_AS uint8_t *)&d + 12);
        const uint32_t live = (uint32_t)k < ci.n ? 0x3fu : 0u;
        ci.cbp[k] = (w3 >> 8) & live;
        ci.flags[k] = (w3 & 0xffu) | (((w3 >> 16) & 0xffu) << 8);
        ci.coef_off[k] = d.coef_off;
        ci.qtab[k] = pics[d.pic].stream * 256 + ((w3 & MPEGHIP_MB_INTRA) ? 0u : 128u);
        ci.base[k] = acc;
        acc += popc6(ci.cbp[k]);
    }
    ci.base[N] = acc;
    return ci;
}

MPG_HD ChunkInfo load_chunk(const VideoArgs &a, uint32_t chunk) { return load_chunk_t<kChunkMbs>(a, chunk); }

// base[w] for a run-time (wave-uniform) w, computed arithmetically: indexing the array
// with a run-time value (or a select chain over its elements, which LLVM folds back into
// an indexed load) would push the whole ChunkInfo into scratch memory.
template <int N>
MPG_HD uint32_t chunk_base_of(const ChunkInfoT<N> &ci, uint32_t w)
{
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++)
        r += ((uint32_t)i < w) ? popc6(ci.cbp[i]) : 0u;
    return r;
}

// ------------------------------------------------------------------ phase 2
// lane (g, j) of a working wave: slot = 8*wave + g.
template <int N>
MPG_HD void compact_phase2(const VideoArgs &a, const ChunkInfoT<N> &ci, uint32_t slot, int j, int32_t *tile_g, bool &active)
{
    active = slot < ci.base[N];
    if (!active)
        return;
    // which macroblock owns this slot: k = #{i >= 1 : base[i] <= slot}
    uint32_t k = 0;
#pragma unroll
    for (int i = 1; i < N; i++)
        k += (slot >= ci.base[i]) ? 1u : 0u;
    uint32_t cbp = ci.cbp[0], flags = ci.flags[0], coef_off = ci.coef_off[0], qtab = ci.qtab[0], base = ci.base[0];
#pragma unroll
    for (int i = 1; i < N; i++) {
        const bool sel = k == (uint32_t)i;
        cbp = sel ? ci.cbp[i] : cbp;
        flags = sel ? ci.flags[i] : flags;
        coef_off = sel ? ci.coef_off[i] : coef_off;
        qtab = sel ? ci.qtab[i] : qtab;
        base = sel ? ci.base[i] : base;
    }
    const uint32_t idx = slot - base; // ordinal among the macroblock's coded blocks
    const bool intra = (flags & MPEGHIP_MB_INTRA) != 0, raw = (flags & MPEGHIP_MB_COEF_RAW) != 0;
    (void)cbp;

    int32_t v[8];
    if (raw) {
        const i32x4 *c = reinterpret_cast<const i32x4 *>(a.coefs + ((uint64_t)coef_off + 2 * idx) * MPEGHIP_COEF_UNIT + (uint32_t)j * 32);
        const i32x4 c0 = c[0], c1 = c[1];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] = c0.v[r];
            v[r + 4] = c1.v[r];
        }
    } else {
        const i32x4 c0 = *reinterpret_cast<const i32x4 *>(a.coefs + ((uint64_t)coef_off + idx) * MPEGHIP_COEF_UNIT + (uint32_t)j * 16);
        const i32x4 tq = *reinterpret_cast<const i32x4 *>(a.qmat + qtab + j * 16);
        const uint64_t qm = (uint64_t)(uint32_t)tq.v[0] | ((uint64_t)(uint32_t)tq.v[1] << 32);
        const uint64_t pm = (uint64_t)(uint32_t)tq.v[2] | ((uint64_t)(uint32_t)tq.v[3] << 32);
        dequant_column(v, c0, qm, pm, (int32_t)((flags >> 8) & 31), intra, j == 0);
    }
    idct8<false>(v);
    int32_t *t = tile_g + j;
#pragma unroll
    for (int r = 0; r < 8; r++)
        t[r * 8] = v[r];
}

// pack two int32 into saturated int16 pair
MPG_HD uint32_t sat_pack_i16(int32_t lo, int32_t hi)
{
#if MPG_ON_DEVICE
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (s16x2)__builtin_amdgcn_cvt_pk_i16(lo, hi));
#else
    const int32_t l = lo < -32768 ? -32768 : (lo > 32767 ? 32767 : lo);
    const int32_t h = hi < -32768 ? -32768 : (hi > 32767 ? 32767 : hi);
    return ((uint32_t)l & 0xffffu) | ((uint32_t)h << 16);
#endif
}

// second half of phase 2 (after the wave-private LDS hand-off): lane (g, j) = row j
MPG_HD void compact_phase2_rows(uint32_t slot, int j, const int32_t *tile_g, bool active, uint8_t *resid_store)
{
    if (!active)
        return;
    int32_t v[8];
    const i32x4 *t = reinterpret_cast<const i32x4 *>(tile_g + j * 8);
    const i32x4 t0 = t[0], t1 = t[1];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        v[c] = t0.v[c];
        v[c + 4] = t1.v[c];
    }
    idct8<true>(v);
    u32x4 row;
    row.v[0] = sat_pack_i16(v[0], v[1]);
    row.v[1] = sat_pack_i16(v[2], v[3]);
    row.v[2] = sat_pack_i16(v[4], v[5]);
    row.v[3] = sat_pack_i16(v[6], v[7]);
    *reinterpret_cast<u32x4 *>(resid_store + slot * 128 + (uint32_t)j * 16) = row;
}

// 4 prediction bytes + two int16 pairs -> 4 clamped bytes
MPG_HD uint32_t add_resid_pack4(uint32_t pred4, uint32_t r01, uint32_t r23)
{
#if MPG_ON_DEVICE
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const uint32_t p01 = __builtin_amdgcn_perm(0u, pred4, 0x0c010c00u);
    const uint32_t p23 = __builtin_amdgcn_perm(0u, pred4, 0x0c030c02u);
    const s16x2 s01 = __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, r01), __builtin_bit_cast(s16x2, p01));
    const s16x2 s23 = __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, r23), __builtin_bit_cast(s16x2, p23));
    uint32_t u01, u23;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(u01) : "v"(__builtin_bit_cast(uint32_t, s01)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(u23) : "v"(__builtin_bit_cast(uint32_t, s23)));
    return __builtin_amdgcn_perm(u23, u01, 0x05040100u);
#else
    const int32_t r[4] = {(int16_t)(r01 & 0xffff), (int16_t)(r01 >> 16), (int16_t)(r23 & 0xffff), (int16_t)(r23 >> 16)};
    uint32_t out = 0;
    for (int c = 0; c < 4; c++) {
        const int32_t x = (int32_t)((pred4 >> (8 * c)) & 0xff) + r[c];
        out |= (uint32_t)(x < 0 ? 0 : (x > 255 ? 255 : x)) << (8 * c);
    }
    return out;
#endif
}

// ------------------------------------------------------------------ phase 3
// wave w (macroblock w of the chunk), lane (b, j): finish row j of block b.
template <int N>
MPG_HD void compact_phase3(const VideoArgs &a, const MbU &u, const ChunkInfoT<N> &ci, uint32_t w, int lane, const MbLoads &ld,
                           const uint8_t *resid_store)
{
    const int b = lane >> 3, j = lane & 7;
    if (b >= 6)
        return;
    const bool intra = (u.flags & MPEGHIP_MB_INTRA) != 0;
    const bool coded = (u.cbp & (0x20u >> b)) != 0;
    if (intra && !coded)
        return; // an invalid intra block leaves the old pixels (video.go:711-714)

    // prediction (video_noasm.go:48-80) from the rows loaded in phase 1
    uint64_t pred = 0;
    if (!intra) {
        int32_t mvx = u.mv_x, mvy = u.mv_y;
        if (b >= 4) {
            mvx /= 2;
            mvy /= 2;
        }
        const bool oh = (mvx & 1) != 0, ov = (mvy & 1) != 0;
        const uint64_t pa = (uint64_t)ld.r0.v[0] | ((uint64_t)ld.r0.v[1] << 32);
        if (!oh && !ov) {
            pred = pa;
        } else {
            const uint64_t pc = (uint64_t)ld.r1.v[0] | ((uint64_t)ld.r1.v[1] << 32);
            if (!oh) {
                pred = avg2_u8x8(pa, pc);
            } else {
                const uint64_t pb = (uint64_t)shift_in_byte(ld.r0.v[1], ld.r0.v[0]) | ((uint64_t)shift_in_byte(ld.r0.v[2], ld.r0.v[1]) << 32);
                if (!ov) {
                    pred = avg2_u8x8(pa, pb);
                } else {
                    const uint64_t pd = (uint64_t)shift_in_byte(ld.r1.v[1], ld.r1.v[0]) | ((uint64_t)shift_in_byte(ld.r1.v[2], ld.r1.v[1]) << 32);
                    pred = avg4_u8x8(pa, pb, pc, pd);
                }
            }
        }
    }
    uint64_t out = pred;
    if (coded) {
        const uint32_t slot = chunk_base_of(ci, w) + popc6(u.cbp >> (6 - b));
        const u32x4 row = *reinterpret_cast<const u32x4 *>(resid_store + slot * 128 + (uint32_t)j * 16);
        const uint32_t lo = add_resid_pack4((uint32_t)pred, row.v[0], row.v[1]);
        const uint32_t hi = add_resid_pack4((uint32_t)(pred >> 32), row.v[2], row.v[3]);
        out = (uint64_t)lo | ((uint64_t)hi << 32);
    }
    uint32_t off;
    if (b < 4) {
        const uint32_t y = (u.mb_y << 4) + (uint32_t)j + ((uint32_t)(b >> 1) << 3);
        const uint32_t x = (u.mb_x << 4) + ((uint32_t)(b & 1) << 3);
        off = y * a.luma_w + x;
    } else {
        const uint32_t y = (u.mb_y << 3) + (uint32_t)j;
        off = a.luma_bytes + (b == 5 ? a.chroma_bytes : 0) + y * a.chroma_w + (u.mb_x << 3);
    }
    *reinterpret_cast<uint64_t *>(u.cur + off) = out;
}

// phase 1 = prediction loads only (no coefficient loads): reuse mb_issue_loads with cbp masked off
MPG_HD void compact_phase1(const VideoArgs &a, const MbU &u, int lane, MbLoads &ld)
{
    MbU v = u;
    v.cbp = 0;
    mb_issue_loads(a, v, lane, ld);
}

// ------------------------------------------------------------------ wave-chunk variant
// The same three phases run by ONE wave on a chunk of kWcMbs macroblocks, sequentially,
// with wave-private LDS: no barrier, no coupling between waves.  All loads of the chunk
// (coefficient columns of the first IDCT pass, prediction rows of every macroblock) are
// issued before the first use.  Two L1-access savers on top (the kernel is bound by L1 line
// accesses, profiles/r01i_pmc_summary.txt):
//   * the row below (vertical half-pel tap) comes from the lane that already loaded it
//     (ds_bpermute) — only the bottom row of the macroblock is loaded a second time;
//   * when the chunk is a horizontal run of 4 macroblocks, the outputs go through LDS and
//     leave as whole 64-byte luma / 32-byte chroma rows (32 stores of 16 bytes per chunk
//     instead of 192 of 8 bytes).
constexpr int kWcMbs = 4;
constexpr int kWcMaxBlocks = 6 * kWcMbs;                   // 24
constexpr int kWcResidBytes = kWcMaxBlocks * 128;           // 3072
// 8 blocks x 64 dwords, no padding: the column writes take 8-way bank conflicts, but the 256 bytes the
// padding would cost per wave are what separates 7 from 8 resident workgroups per CU (20 480 bytes each),
// and the kernel is short of waves, not of LDS bandwidth (4 % of its issue).
constexpr int kWcTileStride = 64;
constexpr int kWcTileBytes = 8 * kWcTileStride * 4;         // 2048: IDCT transpose tile, later the output tile
constexpr int kWcLdsBytes = kWcResidBytes + kWcTileBytes;   // 5376
constexpr int kWcOutBytes = 16 * 64 + 2 * 8 * 32;           // 1536 <= kWcTileBytes
using WcInfo = ChunkInfoT<kWcMbs>;

// The wave-chunk kernel does not read mpeghip_mb_desc / mpeghip_pic_desc.  The HOST half of the library turns
// every macroblock descriptor (+ its picture's) into a 48-byte record while it stages a batch (expand_mb,
// called from the validation loop of mpeghip.hip) and uploads the records; they hold what the kernel
// would otherwise work out per wave on the scalar unit — frame offsets of the block origins and of the
// prediction source, the frames' byte offsets, half-pel flags, table offset.  Why:
//   * the descriptors are streamed (every load misses all caches) and scalar loads return out of
//     order, so each s_waitcnt waits for all of them: descriptor -> picture -> ... cost a wave ten
//     round trips; one round of loads of 4 x 44 bytes is left;
//   * scalar instructions are not free on this path: +20 per macroblock inside phase 3 cost 4 % of the
//     typical workload (more than 20 vector instructions), and two thirds of the kernel's 145 scalar
//     instructions per macroblock were this arithmetic (profiles/r03y_sensitivity_phase3.txt);
//   * with the registers that frees, 8 instead of 7 waves fit a SIMD.
// The C ABI is unchanged: records are the library's device format.  They cost 16 B of descriptor
// traffic per macroblock more than the ABI's 32-byte descriptor, which the HBM has to spare.
constexpr int kXDwords = 12;
constexpr uint32_t kXOhLuma = 1u << 24, kXOvLuma = 1u << 25, kXOhChroma = 1u << 26, kXOvChroma = 1u << 27,
                   kXNeedsBelow = 1u << 28, kXRgba = 1u << 29,
                   kXRun = 1u << 30; // (first record of a chunk) the chunk is a horizontal run: mark_chunk_runs
// dword 0: flags | cbp << 8 | qscale << 16 | kX* bits     1: coef_off      2: qtab (byte offset into qmat)
//       3: mb_x | mb_y << 16     4: cur frame offset >> 8     5: reference frame offset >> 8
//       6..9: src_luma, src_chroma, dst_luma, dst_chroma (bytes inside the frame)   10: RGBA image offset >> 8
// Per stream s of a replicated batch, dwords 1, 2, 4, 5, 10 move by s times kXStep* (replicate_desc_kernel).

// geometry the expansion needs (a subset of VideoArgs, so that host code can call it too)
struct XGeom {
    uint32_t luma_w, chroma_w;
    uint64_t frame_stride, rgba_stride;
};

// the picture's share of a record, worked out once per picture
struct XPic {
    uint32_t cur256, fwd256, bwd256, rgba256, qtab, bits;
};

MPG_HD XPic expand_pic(const XGeom &g, const mpeghip_pic_desc &p)
{
    const uint64_t s3 = (uint64_t)p.stream * MPEGHIP_SLOTS;
    XPic xp;
    xp.cur256 = (uint32_t)(((s3 + p.cur) * g.frame_stride) >> 8); // strides are multiples of 256
    xp.fwd256 = (uint32_t)(((s3 + p.fwd) * g.frame_stride) >> 8);
    xp.bwd256 = (uint32_t)(((s3 + p.bwd) * g.frame_stride) >> 8);
    xp.rgba256 = (uint32_t)(((s3 + p.cur) * g.rgba_stride) >> 8);
    xp.qtab = p.stream * 256;
    xp.bits = (p.flags & MPEGHIP_PIC_RGBA) ? kXRgba : 0;
    return xp;
}

MPG_HD void expand_mb(const XGeom &g, const XPic &xp, const mpeghip_mb_desc &d, uint32_t *x /* [kXDwords] */)
{
    const bool intra = (d.flags & MPEGHIP_MB_INTRA) != 0;
    const int32_t mvx = d.mv_x, mvy = d.mv_y;
    const int32_t cmx = mvx / 2, cmy = mvy / 2; // toward zero, video_noasm.go:35-36
    uint32_t w0 = (uint32_t)d.flags | ((uint32_t)d.cbp << 8) | ((uint32_t)d.qscale << 16) | xp.bits;
    w0 |= (mvx & 1) ? kXOhLuma : 0;
    w0 |= (mvy & 1) ? kXOvLuma : 0;
    w0 |= (cmx & 1) ? kXOhChroma : 0;
    w0 |= (cmy & 1) ? kXOvChroma : 0;
    w0 |= (!intra && ((mvy & 1) || (cmy & 1))) ? kXNeedsBelow : 0;
    const int32_t dst_luma = (int32_t)((uint32_t)d.mb_y << 4) * (int32_t)g.luma_w + (int32_t)((uint32_t)d.mb_x << 4);
    const int32_t dst_chroma = (int32_t)((uint32_t)d.mb_y << 3) * (int32_t)g.chroma_w + (int32_t)((uint32_t)d.mb_x << 3);
    x[0] = w0;
    x[1] = d.coef_off;
    x[2] = xp.qtab + (intra ? 0u : 128u);
    x[3] = (uint32_t)d.mb_x | ((uint32_t)d.mb_y << 16);
    x[4] = xp.cur256;
    x[5] = (d.flags & MPEGHIP_MB_REF_BWD) ? xp.bwd256 : xp.fwd256;
    x[6] = (uint32_t)(dst_luma + (mvy >> 1) * (int32_t)g.luma_w + (mvx >> 1));
    x[7] = (uint32_t)(dst_chroma + (cmy >> 1) * (int32_t)g.chroma_w + (cmx >> 1));
    x[8] = (uint32_t)dst_luma;
    x[9] = (uint32_t)dst_chroma;
    x[10] = xp.rgba256;
    x[11] = 0;
    static_assert(kXDwords == 12, "record layout");
}

// Is chunk c (records 4c .. 4c+3) a horizontal run of 4 fully written macroblocks of one frame, 64-byte aligned?
// Then its outputs leave as whole rows (wc_store_tile).  Worked out on the host for every chunk, after all
// records of the batch are written; kept in the chunk's first record.
MPG_HD void mark_chunk_run(uint32_t *xrec, uint32_t first) // records first .. first+3 exist
{
    uint32_t *x0 = xrec + (size_t)first * kXDwords;
    bool ok = ((x0[3] & 0xffff) & 3) == 0;
    for (int m = 0; m < kWcMbs && ok; m++) {
        const uint32_t *x = x0 + m * kXDwords;
        ok = x[4] == x0[4] && x[3] == x0[3] + (uint32_t)m; // same frame, same row, next column
        const uint32_t flags = x[0] & 0xff, cbp = (x[0] >> 8) & 0xff;
        ok = ok && (!(flags & MPEGHIP_MB_INTRA) || cbp == 0x3f); // an invalid intra block keeps old pixels
    }
    if (ok)
        x0[0] |= kXRun;
}

MPG_HD void mark_chunk_runs(uint32_t *xrec, uint32_t n_mbs)
{
    for (uint32_t first = 0; first + kWcMbs <= n_mbs; first += kWcMbs)
        mark_chunk_run(xrec, first);
}

struct WcRaw {
    uint32_t d[kWcMbs][11];
};

// ONE round of scalar loads for the whole chunk (kRgba = false: the instance for batches without
// colour conversion, which neither loads nor keeps anything of the RGBA images)
template <bool kRgba>
MPG_HD void wc_load_raw(const VideoArgs &a, uint32_t chunk, uint32_t &n, WcRaw &r)
{
    const uint32_t first = chunk * kWcMbs;
    n = a.n_mbs - first < (uint32_t)kWcMbs ? a.n_mbs - first : (uint32_t)kWcMbs;
    const MPG_CONST_AS uint32_t *x = (const MPG_CONST_AS uint32_t *)(uintptr_t)a.xmbs + (uint64_t)first * kXDwords;
#pragma unroll
    for (int k = 0; k < kWcMbs; k++) {
        const uint32_t kk = (uint32_t)k < n ? (uint32_t)k : 0u; // past the end of the batch: macroblock 0 again
#pragma unroll
        for (int w = 0; w < (kRgba ? 11 : 10); w++)
            r.d[k][w] = x[kk * kXDwords + w];
        if (!kRgba)
            r.d[k][10] = 0;
    }
}

// the wave-uniform view the phases use (no arithmetic left beyond two 64-bit adds)
template <bool kRgba>
MPG_HD MbU wc_mb_from_raw(const VideoArgs &a, const uint32_t (&d)[11])
{
    MbU u;
    u.flags = d[0] & 0xff;
    u.cbp = (d[0] >> 8) & 0xff;
    u.qscale = (d[0] >> 16) & 0xff;
    u.bits = d[0];
    u.coef_off = d[1];
    u.mv_x = u.mv_y = 0; // (not used by the wave-chunk phases)
    u.mb_x = d[3] & 0xffff;
    u.mb_y = d[3] >> 16;
    u.pic_flags = (kRgba && (d[0] & kXRgba)) ? MPEGHIP_PIC_RGBA : 0;
    u.cur_off256 = d[4];
    u.cur = a.frames + ((uint64_t)d[4] << 8);
    u.ref = a.frames + ((uint64_t)d[5] << 8);
    u.qm = a.qmat + d[2];
    u.rgba = (kRgba && (d[0] & kXRgba)) ? a.rgba + ((uint64_t)d[10] << 8) : nullptr;
    u.src_luma = (int32_t)d[6];
    u.src_chroma = (int32_t)d[7];
    u.dst_luma = (int32_t)d[8];
    u.dst_chroma = (int32_t)d[9];
    return u;
}

MPG_HD WcInfo wc_info_from_raw(uint32_t n, const WcRaw &r)
{
    WcInfo ci;
    ci.n = n;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < kWcMbs; k++) {
        const uint32_t w0 = r.d[k][0];
        const uint32_t live = (uint32_t)k < n ? 0x3fu : 0u;
        ci.cbp[k] = (w0 >> 8) & live;
        ci.flags[k] = (w0 & 0xffu) | (((w0 >> 16) & 0xffu) << 8);
        ci.coef_off[k] = r.d[k][1];
        ci.qtab[k] = r.d[k][2];
        ci.base[k] = acc;
        acc += popc6(ci.cbp[k]);
    }
    ci.base[kWcMbs] = acc;
    return ci;
}

// lane that holds the row below this lane's row, or -1 if that row must be loaded
// (bottom row of the macroblock: luma row 16, chroma row 8)
MPG_HD int wc_below_lane(int lane)
{
    const int b = lane >> 3, j = lane & 7;
    if (b >= 6)
        return lane;
    if (j < 7)
        return lane + 1;
    if (b < 2)
        return lane + 9; // row 0 of the block underneath (b + 2)
    return -1;
}

// Byte offset of this lane's row inside a frame, for macroblock (0, 0) and vector (0, 0): depends on the lane
// only, so it is computed once per wave and the per-macroblock part stays on the scalar unit.
MPG_HD int32_t wc_lane_row_offset(const VideoArgs &a, int lane)
{
    const int b = lane >> 3, j = lane & 7;
    if (b < 4)
        return (j + ((b >> 1) << 3)) * (int32_t)a.luma_w + ((b & 1) << 3);
    return (int32_t)(a.luma_bytes + (b == 5 ? a.chroma_bytes : 0)) + j * (int32_t)a.chroma_w;
}

// Wave-uniform part of the source offset of macroblock u: luma and chroma variants (scalar arithmetic).
struct WcPredScalars {
    int32_t src_luma, src_chroma;   // frame offset of the block origin + integer part of the vector
    int32_t dst_luma, dst_chroma;   // frame offset of the block origin
    bool oh_luma, ov_luma, oh_chroma, ov_chroma;
};

MPG_HD WcPredScalars wc_pred_scalars(const VideoArgs &, const MbU &u)
{
    WcPredScalars p; // all of it was worked out by expand_mb
    p.dst_luma = u.dst_luma;
    p.dst_chroma = u.dst_chroma;
    p.src_luma = u.src_luma;
    p.src_chroma = u.src_chroma;
    p.oh_luma = (u.bits & kXOhLuma) != 0;
    p.ov_luma = (u.bits & kXOvLuma) != 0;
    p.oh_chroma = (u.bits & kXOhChroma) != 0;
    p.ov_chroma = (u.bits & kXOvChroma) != 0;
    return p;
}

// phase 1: prediction loads of one macroblock (row j of block b; + the row below only where no lane has it)
MPG_HD void wc_issue_pred(const VideoArgs &a, const MbU &u, int lane, MbLoads &ld)
{
    ld.r0 = u8x16{{0, 0, 0, 0}};
    ld.r1 = u8x16{{0, 0, 0, 0}};
    ld.c0 = i32x4{{0, 0, 0, 0}};
    ld.c1 = i32x4{{0, 0, 0, 0}};
    ld.qm = ld.pm = 0;
    const int b = lane >> 3;
    if (b >= 6 || (u.flags & MPEGHIP_MB_INTRA))
        return;
    const WcPredScalars p = wc_pred_scalars(a, u);
    const bool luma = b < 4;
    const int32_t off = wc_lane_row_offset(a, lane) + (luma ? p.src_luma : p.src_chroma);
    const int32_t stride = luma ? (int32_t)a.luma_w : (int32_t)a.chroma_w;
    const uint8_t *src = u.ref + off;
    ld.r0 = ld128u(src);
    if ((luma ? p.ov_luma : p.ov_chroma) && wc_below_lane(lane) < 0)
        ld.r1 = ld128u(src + stride);
}

// does any lane of this macroblock need the row below?  (wave-uniform)
MPG_HD bool wc_needs_below(const MbU &u) { return (u.bits & kXNeedsBelow) != 0; }

// phase 3 of the wave-chunk kernel.  `below` = the 16 bytes of the row under this lane's row, already
// fetched from the owning lane (or from ld.r1 for the bottom rows).  If out_tile != nullptr the 8 output
// bytes are parked there (layout: luma [16 rows][4 macroblocks x 16 B], Cb [8][4 x 8 B], Cr [8][4 x 8 B]);
// with `store` they go to the frame as one 8-byte store (both when the picture is colour-converted
// from the tile but the chunk is not a horizontal run).  A block the macroblock does not write (an
// invalid intra block) parks the frame's current bytes, so that the tile always mirrors the planes.
MPG_HD uint32_t wc_tile_offset(int b, int j, uint32_t m)
{
    if (b < 4)
        return ((uint32_t)j + ((uint32_t)(b >> 1) << 3)) * 64 + m * 16 + ((uint32_t)(b & 1) << 3);
    return 1024 + (uint32_t)(b - 4) * 256 + (uint32_t)j * 32 + m * 8;
}

// kRun: the chunk is a horizontal run (wc_can_coalesce) — every block is written, the outputs go to the tile
// only; the instance without the rare paths.
template <int N, bool kRun>
MPG_HD void wc_phase3(const VideoArgs &a, const MbU &u, const ChunkInfoT<N> &ci, uint32_t m, int lane, const MbLoads &ld,
                      const u8x16 &below, const uint8_t *resid_store, uint8_t *out_tile, bool store)
{
    const int b = lane >> 3, j = lane & 7;
    if (b >= 6)
        return;
    const bool intra = (u.flags & MPEGHIP_MB_INTRA) != 0;
    const bool coded = (u.cbp & (0x20u >> b)) != 0;
    if (!kRun && intra && !coded) { // an invalid intra block leaves the old pixels (video.go:711-714); never on the coalesced path
        if (out_tile) {
            const WcPredScalars pk = wc_pred_scalars(a, u);
            const int32_t off = wc_lane_row_offset(a, lane) + (b < 4 ? pk.dst_luma : pk.dst_chroma);
            *reinterpret_cast<uint64_t *>(out_tile + wc_tile_offset(b, j, m)) = *reinterpret_cast<const uint64_t *>(u.cur + off);
        }
        return;
    }

    const WcPredScalars ps = wc_pred_scalars(a, u);
    const bool luma = b < 4;
    uint64_t pred = 0;
    if (!intra) { // video_noasm.go:48-80
        const bool oh = luma ? ps.oh_luma : ps.oh_chroma, ov = luma ? ps.ov_luma : ps.ov_chroma;
        const uint64_t pa = (uint64_t)ld.r0.v[0] | ((uint64_t)ld.r0.v[1] << 32);
        if (!oh && !ov) {
            pred = pa;
        } else {
            const uint64_t pc = (uint64_t)below.v[0] | ((uint64_t)below.v[1] << 32);
            if (!oh) {
                pred = avg2_u8x8(pa, pc);
            } else {
                const uint64_t pb = (uint64_t)shift_in_byte(ld.r0.v[1], ld.r0.v[0]) | ((uint64_t)shift_in_byte(ld.r0.v[2], ld.r0.v[1]) << 32);
                if (!ov) {
                    pred = avg2_u8x8(pa, pb);
                } else {
                    const uint64_t pd = (uint64_t)shift_in_byte(below.v[1], below.v[0]) | ((uint64_t)shift_in_byte(below.v[2], below.v[1]) << 32);
                    pred = avg4_u8x8(pa, pb, pc, pd);
                }
            }
        }
    }
    uint64_t out = pred;
    if (coded) {
        const uint32_t slot = chunk_base_of(ci, m) + popc6(u.cbp >> (6 - b));
        const u32x4 row = *reinterpret_cast<const u32x4 *>(resid_store + slot * 128 + (uint32_t)j * 16);
        const uint32_t lo = add_resid_pack4((uint32_t)pred, row.v[0], row.v[1]);
        const uint32_t hi = add_resid_pack4((uint32_t)(pred >> 32), row.v[2], row.v[3]);
        out = (uint64_t)lo | ((uint64_t)hi << 32);
    }
    if (kRun || out_tile)
        *reinterpret_cast<uint64_t *>(out_tile + wc_tile_offset(b, j, m)) = out;
    if (!kRun && store) {
        const int32_t off = wc_lane_row_offset(a, lane) + (luma ? ps.dst_luma : ps.dst_chroma);
        *reinterpret_cast<uint64_t *>(u.cur + off) = out;
    }
}

// Frame.RGBA fused into the reconstruction (pictures flagged MPEGHIP_PIC_RGBA): macroblock m of the
// chunk from the output tile, 4 pixels per lane (lane = row*4 + segment), one 16-byte store each — a
// macroblock row is 64 contiguous bytes of the image.  Pixels outside width x height are not stored.
MPG_HD void wc_rgba_mb(const VideoArgs &a, const MbU &u, uint32_t m, int lane, const uint8_t *out_tile)
{
    const uint32_t row = (uint32_t)lane >> 2, seg = (uint32_t)lane & 3;
    const uint32_t py = (u.mb_y << 4) + row, px0 = (u.mb_x << 4) + seg * 4;
    if (py >= a.height || px0 >= a.width)
        return;
    const uint32_t yy = *reinterpret_cast<const uint32_t *>(out_tile + row * 64 + m * 16 + seg * 4);
    const uint32_t cb = *reinterpret_cast<const uint16_t *>(out_tile + 1024 + (row >> 1) * 32 + m * 8 + seg * 2);
    const uint32_t cr = *reinterpret_cast<const uint16_t *>(out_tile + 1280 + (row >> 1) * 32 + m * 8 + seg * 2);
    uint32_t px[4];
    rgba_row4(yy, chroma_terms(cb & 0xff, cr & 0xff), chroma_terms((cb >> 8) & 0xff, (cr >> 8) & 0xff), px);
    const uint64_t p = (uint64_t)py * a.width + px0;
    const uint32_t n = a.width - px0 >= 4 ? 4 : a.width - px0;
    rgba_store4<false>(reinterpret_cast<uint32_t *>(u.rgba) + p, p, px, n);
}

// Is the chunk a horizontal run of 4 fully written macroblocks of one picture, 64-byte aligned?  (wave-uniform;
// decided by the host: mark_chunk_runs)
MPG_HD bool wc_can_coalesce(const WcInfo &ci, const MbU (&u)[kWcMbs])
{
    return ci.n == (uint32_t)kWcMbs && (u[0].bits & kXRun) != 0;
}

// Cooperative store of the chunk's output tile: luma 16 rows x 64 B by all 64 lanes, chroma 2 x 8 rows x 32 B by lanes 0-31.
MPG_HD void wc_store_tile(const VideoArgs &a, const MbU &u0, int lane, const uint8_t *out_tile)
{
    {
        const uint32_t row = (uint32_t)lane >> 2, seg = (uint32_t)lane & 3;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(out_tile + row * 64 + seg * 16);
        *reinterpret_cast<u32x4 *>(u0.cur + ((u0.mb_y << 4) + row) * a.luma_w + (u0.mb_x << 4) + seg * 16) = v;
    }
    if (lane < 32) {
        const uint32_t plane = (uint32_t)lane >> 4, row = ((uint32_t)lane >> 1) & 7, seg = (uint32_t)lane & 1;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(out_tile + 1024 + plane * 256 + row * 32 + seg * 16);
        *reinterpret_cast<u32x4 *>(u0.cur + a.luma_bytes + plane * a.chroma_bytes + ((u0.mb_y << 3) + row) * a.chroma_w +
                                   (u0.mb_x << 3) + seg * 16) = v;
    }
}

} // namespace mpg
