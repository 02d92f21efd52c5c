// video_split_lane.h — the reconstruction path as two dense kernels.
//
// Measured on MI355X (profiles/r01b_*): the fused one-wave-per-macroblock kernel is
// VALU-issue bound (84 % VALU busy at 22 % of the HBM roofline) because every
// macroblock-wave executes the whole dequantise + IDCT instruction stream even
// when only one or two of its six blocks are coded (mean 1.84 in real streams).
// Splitting the work by what is actually dense fixes that:
//
//   K1 pred_kernel   one wave = TWO macroblocks, 32 lanes each, every lane busy:
//                    lanes 0-15  luma row t, 16 bytes (one unaligned 16-byte load per tap)
//                    lanes 16-23 Cb row, lanes 24-31 Cr row, 8 bytes each
//                    predictMacroblock/copyMacroblock (video.go:608-637,
//                    video_noasm.go:28-80).  Lanes 0-5 of each half also write the
//                    work-list entry of their (coded) block for K2.
//   K2 resid_kernel  one wave = EIGHT coded blocks, lane (g, j) = column j of block g:
//                    dequantise + premultiply (video.go:719-744), column pass, LDS
//                    transpose, row pass (video.go:801-928), then add to / overwrite the
//                    destination (video.go:943-1002).  Work items are the 128-byte units
//                    of the coefficient stream, which is already a dense list of coded
//                    blocks.
//
// K2 re-reads 64 destination bytes per inter coded block that K1 wrote; that is the
// price of density.  Results are bit-identical to the fused kernel.
#pragma once

#include "video_lane.h"

namespace mpg {

// One entry per 128-byte coefficient unit, written by K1, read by K2.
struct alignas(16) BlockEntry {
    uint32_t dest_lo;   // byte offset of the block's first pixel from VideoArgs::frames
    uint32_t dest_hi;   // bits 0-15 offset high; bit 16 intra, bit 17 raw, bit 18 chroma; bits 24-28 quantiser_scale
    uint32_t qtable;    // byte offset of the {matrix, premultiplier} column table
    uint32_t skip;      // 0 = valid, ~0 = not a block start (second half of an int32 block / unused unit)
};

constexpr uint32_t kEntryIntra = 1u << 16, kEntryRaw = 1u << 17, kEntryChroma = 1u << 18;

struct SplitArgs {
    VideoArgs v;
    BlockEntry *entries;   // [n_units]
    uint32_t n_units;      // coefficient stream length in 128-byte units
};

// ------------------------------------------------------------------------- K1
// Everything K1 needs to know about one macroblock, resolved from the two
// descriptors.  On the device it is produced by SCALAR loads through the constant
// address space (one set per half-wave, then a per-lane select): the descriptor and
// picture fetches stay off the vector-memory path, which shortens the dependent
// chain descriptor -> picture -> reference pixels -> store that bounds this kernel.
struct PredMb {
    uint32_t flags, cbp, qscale, coef_off;
    int32_t mv_x, mv_y;
    uint32_t mb_x, mb_y;
    uint32_t stream;
    uint32_t cur_slot, ref_slot;
};

MPG_HD PredMb load_pred_mb(const VideoArgs &a, uint32_t mb_index)
{
    const MPG_CONST_AS mpeghip_mb_desc *mbs = (const MPG_CONST_AS mpeghip_mb_desc *)(uintptr_t)a.mbs;
    const MPG_CONST_AS mpeghip_pic_desc *pics = (const MPG_CONST_AS mpeghip_pic_desc *)(uintptr_t)a.pics;
    const MPG_CONST_AS mpeghip_mb_desc &d = mbs[mb_index];
    const MPG_CONST_AS mpeghip_pic_desc &p = pics[d.pic];
    const uint32_t slots = *(const MPG_CONST_AS uint32_t *)((const MPG_CONST_AS uint8_t *)&p + 4);
    PredMb m;
    m.flags = d.flags;
    m.cbp = d.cbp;
    m.qscale = d.qscale;
    m.coef_off = d.coef_off;
    m.mv_x = d.mv_x;
    m.mv_y = d.mv_y;
    m.mb_x = d.mb_x;
    m.mb_y = d.mb_y;
    m.stream = p.stream;
    m.cur_slot = slots & 0xff;
    m.ref_slot = (d.flags & MPEGHIP_MB_REF_BWD) ? (slots >> 16) & 0xff : (slots >> 8) & 0xff;
    return m;
}

MPG_HD PredMb select_pred_mb(bool second, const PredMb &x, const PredMb &y)
{
    PredMb m;
    m.flags = second ? y.flags : x.flags;
    m.cbp = second ? y.cbp : x.cbp;
    m.qscale = second ? y.qscale : x.qscale;
    m.coef_off = second ? y.coef_off : x.coef_off;
    m.mv_x = second ? y.mv_x : x.mv_x;
    m.mv_y = second ? y.mv_y : x.mv_y;
    m.mb_x = second ? y.mb_x : x.mb_x;
    m.mb_y = second ? y.mb_y : x.mb_y;
    m.stream = second ? y.stream : x.stream;
    m.cur_slot = second ? y.cur_slot : x.cur_slot;
    m.ref_slot = second ? y.ref_slot : x.ref_slot;
    return m;
}

// One lane of K1: t = lane & 31 within the macroblock's half-wave.
MPG_HD void pred_lane(const SplitArgs &s, const PredMb &d, int t)
{
    const VideoArgs &a = s.v;
    const bool intra = (d.flags & MPEGHIP_MB_INTRA) != 0;
    const uint64_t s3 = (uint64_t)d.stream * MPEGHIP_SLOTS;
    const uint64_t cur_off = (s3 + d.cur_slot) * a.frame_stride;

    // ---- work-list entries for K2 (lanes 0..5 <-> blocks 0..5)
    if (t < 6 && (d.cbp & (0x20u >> t))) {
        const int b = t;
        const bool raw = (d.flags & MPEGHIP_MB_COEF_RAW) != 0;
        const uint32_t k = popc6(d.cbp >> (6 - b));
        const uint32_t unit = d.coef_off + (raw ? 2 * k : k);
        uint64_t off;
        if (b < 4)
            off = (uint64_t)((d.mb_y << 4) + ((uint32_t)(b >> 1) << 3)) * a.luma_w + (d.mb_x << 4) + ((uint32_t)(b & 1) << 3);
        else
            off = (uint64_t)a.luma_bytes + (b == 5 ? a.chroma_bytes : 0) + (uint64_t)(d.mb_y << 3) * a.chroma_w + (d.mb_x << 3);
        off += cur_off;
        BlockEntry e;
        e.dest_lo = (uint32_t)off;
        e.dest_hi = (uint32_t)(off >> 32) | (intra ? kEntryIntra : 0) | (raw ? kEntryRaw : 0) | (b >= 4 ? kEntryChroma : 0) |
                    (d.qscale << 24);
        e.qtable = d.stream * 256 + (intra ? 0 : 128);
        e.skip = 0;
        if (unit < s.n_units)
            s.entries[unit] = e;
        if (raw && unit + 1 < s.n_units) { // second half of an int32 block is not a block start
            e.skip = ~0u;
            s.entries[unit + 1] = e;
        }
    }
    if (intra)
        return;

    // ---- prediction
    const uint8_t *ref = a.frames + (s3 + d.ref_slot) * a.frame_stride;
    uint8_t *cur = a.frames + cur_off;
    int32_t mvx = d.mv_x, mvy = d.mv_y;
    if (t < 16) { // luma row t: 16 pixels
        const int32_t stride = (int32_t)a.luma_w;
        const int32_t y = (int32_t)(d.mb_y << 4) + t;
        const int32_t x = (int32_t)(d.mb_x << 4);
        const uint8_t *src = ref + (y + (mvy >> 1)) * stride + x + (mvx >> 1);
        const bool oh = (mvx & 1) != 0, ov = (mvy & 1) != 0;
        u8x16 o = ld128u(src);
        if (oh && ov) {
            const u8x16 b1 = ld128u(src + 1), c1 = ld128u(src + stride), d1 = ld128u(src + stride + 1);
#pragma unroll
            for (int k = 0; k < 4; k++)
                o.v[k] = avg4_u8x4(o.v[k], b1.v[k], c1.v[k], d1.v[k]);
        } else if (oh || ov) {
            const u8x16 b1 = ld128u(src + (oh ? 1 : stride));
#pragma unroll
            for (int k = 0; k < 4; k++)
                o.v[k] = avg_ceil_u8x4(o.v[k], b1.v[k]);
        }
        *reinterpret_cast<u8x16 *>(cur + y * stride + x) = o;
    } else { // chroma: t 16..23 Cb rows, 24..31 Cr rows: 8 pixels
        mvx /= 2; // toward zero, video_noasm.go:35-36
        mvy /= 2;
        const int32_t stride = (int32_t)a.chroma_w;
        const int32_t r = (t - 16) & 7;
        const uint32_t plane = a.luma_bytes + (t >= 24 ? a.chroma_bytes : 0);
        const int32_t y = (int32_t)(d.mb_y << 3) + r;
        const int32_t x = (int32_t)(d.mb_x << 3);
        const uint8_t *src = ref + plane + (y + (mvy >> 1)) * stride + x + (mvx >> 1);
        const bool oh = (mvx & 1) != 0, ov = (mvy & 1) != 0;
        uint64_t o = ld64u(src);
        if (oh && ov)
            o = avg4_u8x8(o, ld64u(src + 1), ld64u(src + stride), ld64u(src + stride + 1));
        else if (oh || ov)
            o = avg2_u8x8(o, ld64u(src + (oh ? 1 : stride)));
        *reinterpret_cast<uint64_t *>(cur + plane + y * stride + x) = o;
    }
}

// ------------------------------------------------------------------------- K2
constexpr int kResidTileDwords = 8 * kTileStride; // 8 blocks per wave

struct ResidLane {
    uint64_t dest;     // byte offset of this lane's ROW (row j) from frames, or ~0 if idle
    uint64_t pred;     // the 8 destination bytes (inter) / 0 (intra)
    bool active;
};

// phase A of K2: lane (g, j) = column j of the wave's g-th unit.
MPG_HD void resid_phase_a(const SplitArgs &s, uint32_t unit, int j, int32_t *tile_g, ResidLane &st)
{
    const VideoArgs &a = s.v;
    st.active = false;
    st.pred = 0;
    st.dest = 0;
    if (unit >= s.n_units)
        return;
    const BlockEntry e = s.entries[unit];
    if (e.skip != 0)
        return;
    st.active = true;
    const bool intra = (e.dest_hi & kEntryIntra) != 0, raw = (e.dest_hi & kEntryRaw) != 0;
    const uint32_t stride = (e.dest_hi & kEntryChroma) ? a.chroma_w : a.luma_w;
    st.dest = ((uint64_t)(e.dest_hi & 0xffff) << 32 | e.dest_lo) + (uint64_t)j * stride;
    if (!intra)
        st.pred = *reinterpret_cast<const uint64_t *>(a.frames + st.dest); // written by K1

    int32_t v[8];
    const uint8_t *cp = a.coefs + (uint64_t)unit * MPEGHIP_COEF_UNIT;
    if (raw) {
        const i32x4 *c = reinterpret_cast<const i32x4 *>(cp + (uint32_t)j * 32);
        const i32x4 c0 = c[0], c1 = c[1];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] = c0.v[r];
            v[r + 4] = c1.v[r];
        }
    } else {
        const i32x4 c0 = *reinterpret_cast<const i32x4 *>(cp + (uint32_t)j * 16);
        const i32x4 tq = *reinterpret_cast<const i32x4 *>(a.qmat + e.qtable + j * 16);
        const uint64_t qm = (uint64_t)(uint32_t)tq.v[0] | ((uint64_t)(uint32_t)tq.v[1] << 32);
        const uint64_t pm = (uint64_t)(uint32_t)tq.v[2] | ((uint64_t)(uint32_t)tq.v[3] << 32);
        dequant_column(v, c0, qm, pm, (int32_t)((e.dest_hi >> 24) & 31), intra, j == 0);
    }
    idct8<false>(v);
    int32_t *t = tile_g + j;
#pragma unroll
    for (int r = 0; r < 8; r++)
        t[r * 8] = v[r];
}

// phase B of K2: lane (g, j) = row j of the same block.
MPG_HD void resid_phase_b(const SplitArgs &s, int j, const int32_t *tile_g, const ResidLane &st)
{
    if (!st.active)
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
    *reinterpret_cast<uint64_t *>(s.v.frames + st.dest) = add_clamp_pack8(st.pred, v);
}

} // namespace mpg
