// video_lane.h — one wavefront reconstructs one macroblock.
//
// Replaces, per macroblock: predictMacroblock/copyMacroblock (video.go:608-637,
// video_noasm.go:28-80, video_amd64.s, video_arm64.s), the dequantise +
// premultiply tail of decodeBlock (video.go:719-744), idct (video.go:801-928)
// and copy/add*ToDest (video.go:943-1002); optionally Frame.RGBA (video.go:31-36).
//
// Lane map (wave64): lane = b*8 + j, b = block 0..5 (0-3 luma raster, 4 Cb, 5 Cr),
// lanes 48..63 idle.
//   phase A: lane (b,j) owns COLUMN j of block b: loads its 8 coefficients (one
//            16-byte load, coefficient blocks are stored column-major), dequantises,
//            runs the column pass, parks the 8 results in the wave's LDS tile.
//            The same lane, acting as ROW j of block b, fetches its 8 prediction
//            pixels straight from the reference frame (unaligned 8-byte loads).
//   phase B: lane (b,j) owns ROW j of block b: reads the row back from LDS, runs
//            the row pass (+128>>8), adds the prediction, clamps, and stores its
//            8 output bytes with one 8-byte store.
//   phase C: (only for pictures flagged MPEGHIP_PIC_RGBA) the 384 output bytes go
//            through LDS once more and 32 lanes colour-convert 8 pixels each.
#pragma once

#include "lane_common.h"
#include "mpeghip.h"

namespace mpg {

struct VideoArgs {
    uint8_t *frames;              // base of the frame store
    uint64_t frame_stride;        // bytes between consecutive (stream, slot) frames
    uint32_t luma_w, luma_h;      // padded plane sizes
    uint32_t chroma_w, chroma_h;
    uint32_t luma_bytes, chroma_bytes;
    const mpeghip_pic_desc *pics;
    const mpeghip_mb_desc *mbs;
    const uint8_t *coefs;         // 128-byte units
    const uint8_t *qmat;          // [n_streams][2 classes][8 columns][16]: per column 8 quantiser-matrix
                                  // bytes (rows 0-7) then the 8 premultiplier bytes of that column
    uint8_t *dump;                // scratch: 512 bytes per resident wave (sink of the static-count stores)
    uint32_t n_mbs;
    uint32_t width, height;       // display size (RGBA image)
    uint8_t *rgba;                // base of RGBA images, same (stream, slot) indexing
    uint64_t rgba_stride;
    const uint32_t *xmbs;         // wave-chunk kernel: 48-byte expanded macroblock records (video_compact_lane.h)
};

constexpr int kTileStride = 72;               // dwords per 8x8 block in LDS (64 + 8 pad)
constexpr int kTileDwords = 6 * kTileStride;  // per wave
constexpr int kRgbaBytes = 384;               // per wave, phase C staging

// Wave-uniform view of one macroblock descriptor.
struct MbU {
    uint32_t flags, cbp, qscale, coef_off;
    int32_t mv_x, mv_y;
    uint32_t mb_x, mb_y;
    uint32_t pic_flags;
    uint8_t *cur;
    const uint8_t *ref;
    const uint8_t *qm;   // 128-byte column table {matrix column, premultiplier column} of this macroblock's class
    uint8_t *rgba;       // RGBA image of the cur slot (or nullptr)
    // wave-chunk kernel only (from the expanded record): frame byte offsets of the block origins / of the
    // prediction source, half-pel flags
    int32_t src_luma, src_chroma, dst_luma, dst_chroma;
    uint32_t bits;       // kXOhLuma ...
    uint32_t cur_off256; // byte offset of the destination frame >> 8 (names the frame)
};

// Descriptors are read-only for the whole launch.  On the device they are read
// through the constant address space so that the compiler keeps them as scalar
// (s_load) instructions even inside the persistent loop, after global stores.
#if MPG_ON_DEVICE
#define MPG_CONST_AS __attribute__((address_space(4)))
#else
#define MPG_CONST_AS
#endif

MPG_HD MbU load_mb(const VideoArgs &a, uint32_t mb_index)
{
    const MPG_CONST_AS mpeghip_mb_desc *mbs = (const MPG_CONST_AS mpeghip_mb_desc *)(uintptr_t)a.mbs;
    const MPG_CONST_AS mpeghip_pic_desc *pics = (const MPG_CONST_AS mpeghip_pic_desc *)(uintptr_t)a.pics;
    const MPG_CONST_AS mpeghip_mb_desc &d = mbs[mb_index];
    const MPG_CONST_AS mpeghip_pic_desc &p = pics[d.pic];
    MbU u;
    u.flags = d.flags;
    u.cbp = d.cbp;
    u.qscale = d.qscale;
    u.coef_off = d.coef_off;
    u.mv_x = d.mv_x;
    u.mv_y = d.mv_y;
    u.mb_x = d.mb_x;
    u.mb_y = d.mb_y;
    // cur | fwd<<8 | bwd<<16 | flags<<24 in one (scalar) dword load
    const uint32_t slots = *(const MPG_CONST_AS uint32_t *)((const MPG_CONST_AS uint8_t *)&p + 4);
    const uint32_t cur_slot = slots & 0xff;
    u.pic_flags = slots >> 24;
    const uint64_t s3 = (uint64_t)p.stream * MPEGHIP_SLOTS;
    u.cur = a.frames + (s3 + cur_slot) * a.frame_stride;
    const uint32_t ref_slot = (d.flags & MPEGHIP_MB_REF_BWD) ? (slots >> 16) & 0xff : (slots >> 8) & 0xff;
    u.ref = a.frames + (s3 + ref_slot) * a.frame_stride;
    u.qm = a.qmat + (uint64_t)p.stream * 256 + ((d.flags & MPEGHIP_MB_INTRA) ? 0 : 128);
    u.rgba = (u.pic_flags & MPEGHIP_PIC_RGBA) ? a.rgba + (s3 + cur_slot) * a.rgba_stride : nullptr;
    return u;
}

// One 8-point pass of the reference IDCT (video.go:870-895 column form,
// :900-925 row form with the final +128>>8).  int32 is sufficient: every product
// operand stays below 2^23 and every sum below 2^31 for any coefficient block the
// dequantiser can produce (asserted in the emulator build).
template <bool kFinalShift>
MPG_HD void idct8(int32_t (&v)[8])
{
    const int32_t b1 = v[4];
    const int32_t b3 = v[2] + v[6];
    const int32_t b4 = v[5] - v[3];
    const int32_t tmp1 = v[1] + v[7];
    const int32_t tmp2 = v[3] + v[5];
    const int32_t b6 = v[1] - v[7];
    const int32_t b7 = tmp1 + tmp2;
    const int32_t m0 = v[0];
    const int32_t x4 = ((mul24(b6, 473) - mul24(b4, 196) + 128) >> 8) - b7;
    const int32_t x0 = x4 - ((mul24(tmp1 - tmp2, 362) + 128) >> 8);
    const int32_t x1 = m0 - b1;
    const int32_t x2 = ((mul24(v[2] - v[6], 362) + 128) >> 8) - b3;
    const int32_t x3 = m0 + b1;
    const int32_t y3 = x1 + x2;
    const int32_t y4 = x3 + b3;
    const int32_t y5 = x1 - x2;
    const int32_t y6 = x3 - b3;
    const int32_t y7 = -x0 - ((mul24(b4, 473) + mul24(b6, 196) + 128) >> 8);
    v[0] = b7 + y4;
    v[1] = x4 + y3;
    v[2] = y5 - x0;
    v[3] = y6 - y7;
    v[4] = y6 + y7;
    v[5] = x0 + y5;
    v[6] = y3 - x4;
    v[7] = y4 - b7;
    if (kFinalShift) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            v[k] = (v[k] + 128) >> 8;
    }
}

// Dequantise one quantised level the way the VLC loop does (video.go:719-741),
// then premultiply (video.go:744).  q != 0.  qsqm = quantiser_scale * matrix entry.
MPG_HD int32_t dequant(int32_t q, bool intra, int32_t qsqm, int32_t pm)
{
    // level = 2q (+ sign(q) unless intra); q != 0 so sign(q) = (q >> 31) | 1
    int32_t l = 2 * q;
    if (!intra)
        l += (q >> 31) | 1;
    l = mul24(l, qsqm) >> 4;          // |l| <= 513, qsqm <= 31*255
    // "if even, move one toward zero; 0 becomes +1" == (l - (l > 0)) | 1
    l = (l - (l > 0 ? 1 : 0)) | 1;
    l = clampi(l, -2048, 2047);
    return mul24(l, pm);
}

// the 8 quantised levels of one coefficient column -> dequantised, premultiplied
MPG_HD void dequant_column(int32_t (&v)[8], const i32x4 &c0, uint64_t qm, uint64_t pm, int32_t qs, bool intra, bool dc_lane)
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int32_t w = c0.v[r >> 1];
        const int32_t q = (r & 1) ? (w >> 16) : (int32_t)(int16_t)(w & 0xffff);
        if (none_in_wave(q != 0)) { // most rows of most blocks are empty: skip them wave-wide
            v[r] = 0;
            continue;
        }
        const int32_t qsqm = qs * (int32_t)((qm >> (8 * r)) & 0xff);
        const int32_t p = (int32_t)((pm >> (8 * r)) & 0xff);
        const int32_t d = dequant(q, intra, qsqm, p);
        v[r] = q ? d : 0;
    }
    if (intra && dc_lane)
        v[0] = (int32_t)(int16_t)(c0.v[0] & 0xffff) * 256; // DC: `<<= 3+5`, video.go:672
}

struct MbLane {
    uint64_t pred;   // 8 prediction pixels of this lane's row
};

MPG_HD uint32_t popc6(uint32_t x) { return (uint32_t)__builtin_popcount(x & 0x3f); }

// ------------------------------------------------------------------ phase A
// Phase A is split in two so that a persistent wave can have the loads of the
// NEXT macroblock in flight while it computes the current one:
//   mb_issue_loads  — every global load the macroblock needs (prediction source
//                     qwords, coefficient column, quantiser-matrix column), no use
//   mb_phase_a_compute — averages, dequantisation, column pass, LDS tile write
struct MbLoads {
    u8x16 r0, r1;            // prediction source rows: 16 bytes from (row, x) and from (row+1, x); the lane
                             // needs bytes 0..8 of each — ONE load per row serves both horizontal taps
                             // (the reference reads the same 9 bytes as two overlapping 8-byte words)
    i32x4 c0, c1;            // coefficient column: int16 x8 in c0, or int32 x8 in c0,c1
    uint64_t qm;             // quantiser matrix column (8 bytes)
    uint64_t pm;             // premultiplier column (8 bytes)
};

MPG_HD void mb_issue_loads(const VideoArgs &a, const MbU &u, int lane, MbLoads &ld)
{
    const int b = lane >> 3, j = lane & 7;
    ld.r0 = u8x16{{0, 0, 0, 0}};
    ld.r1 = u8x16{{0, 0, 0, 0}};
    ld.qm = ld.pm = 0;
    ld.c0 = i32x4{{0, 0, 0, 0}};
    ld.c1 = i32x4{{0, 0, 0, 0}};
    if (b >= 6)
        return;
    const bool intra = (u.flags & MPEGHIP_MB_INTRA) != 0;
    if (!intra) {
        int32_t mvx = u.mv_x, mvy = u.mv_y;
        int32_t stride, off;
        if (b < 4) {
            stride = (int32_t)a.luma_w;
            const int32_t y = (int32_t)(u.mb_y << 4) + j + ((b >> 1) << 3) + (mvy >> 1);
            const int32_t x = (int32_t)(u.mb_x << 4) + ((b & 1) << 3) + (mvx >> 1);
            off = y * stride + x;
        } else {
            mvx /= 2; // toward zero, video_noasm.go:35-36
            mvy /= 2;
            stride = (int32_t)a.chroma_w;
            const int32_t y = (int32_t)(u.mb_y << 3) + j + (mvy >> 1);
            const int32_t x = (int32_t)(u.mb_x << 3) + (mvx >> 1);
            off = (int32_t)(a.luma_bytes + (b == 5 ? a.chroma_bytes : 0)) + y * stride + x;
        }
        const uint8_t *src = u.ref + off;
        ld.r0 = ld128u(src);
        if (mvy & 1)
            ld.r1 = ld128u(src + stride);
    }
    if (!(u.cbp & (0x20u >> b)))
        return;
    const uint32_t k = popc6(u.cbp >> (6 - b)); // coded blocks before b
    if (u.flags & MPEGHIP_MB_COEF_RAW) {
        const i32x4 *c = reinterpret_cast<const i32x4 *>(
            a.coefs + ((uint64_t)u.coef_off + 2 * k) * MPEGHIP_COEF_UNIT + (uint32_t)j * 32);
        ld.c0 = c[0];
        ld.c1 = c[1];
    } else {
        ld.c0 = *reinterpret_cast<const i32x4 *>(
            a.coefs + ((uint64_t)u.coef_off + k) * MPEGHIP_COEF_UNIT + (uint32_t)j * 16);
        const i32x4 t = *reinterpret_cast<const i32x4 *>(u.qm + j * 16);
        ld.qm = (uint64_t)(uint32_t)t.v[0] | ((uint64_t)(uint32_t)t.v[1] << 32);
        ld.pm = (uint64_t)(uint32_t)t.v[2] | ((uint64_t)(uint32_t)t.v[3] << 32);
    }
}

MPG_HD void mb_phase_a_compute(const VideoArgs &a, const MbU &u, int lane, const MbLoads &ld, MbLane &st, int32_t *tile)
{
    (void)a;
    st.pred = 0;
    const int b = lane >> 3, j = lane & 7;
    if (b >= 6)
        return;
    const bool intra = (u.flags & MPEGHIP_MB_INTRA) != 0;

    // ---- prediction: row j of block b (video_noasm.go:48-80)
    if (!intra) {
        int32_t mvx = u.mv_x, mvy = u.mv_y;
        if (b >= 4) {
            mvx /= 2;
            mvy /= 2;
        }
        const bool oh = (mvx & 1) != 0, ov = (mvy & 1) != 0;
        const uint64_t pa = (uint64_t)ld.r0.v[0] | ((uint64_t)ld.r0.v[1] << 32);
        if (!oh && !ov) {
            st.pred = pa;
        } else {
            const uint64_t pc = (uint64_t)ld.r1.v[0] | ((uint64_t)ld.r1.v[1] << 32);
            if (!oh) {
                st.pred = avg2_u8x8(pa, pc);
            } else {
                const uint64_t pb = (uint64_t)shift_in_byte(ld.r0.v[1], ld.r0.v[0]) | ((uint64_t)shift_in_byte(ld.r0.v[2], ld.r0.v[1]) << 32);
                if (!ov) {
                    st.pred = avg2_u8x8(pa, pb);
                } else {
                    const uint64_t pd = (uint64_t)shift_in_byte(ld.r1.v[1], ld.r1.v[0]) | ((uint64_t)shift_in_byte(ld.r1.v[2], ld.r1.v[1]) << 32);
                    st.pred = avg4_u8x8(pa, pb, pc, pd);
                }
            }
        }
    }

    // ---- residual: column j of block b
    if (!(u.cbp & (0x20u >> b)))
        return;
    int32_t v[8];
    if (u.flags & MPEGHIP_MB_COEF_RAW) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] = ld.c0.v[r];
            v[r + 4] = ld.c1.v[r];
        }
    } else {
        dequant_column(v, ld.c0, ld.qm, ld.pm, (int32_t)u.qscale, intra, j == 0);
    }
    idct8<false>(v);
    int32_t *t = tile + b * kTileStride + j;
#pragma unroll
    for (int r = 0; r < 8; r++)
        t[r * 8] = v[r];
}

// Branch-free variant of mb_issue_loads for the software pipeline: EVERY lane issues
// exactly four loads (2 x 16 bytes of prediction source, 2 x 16 bytes of coefficients /
// tables) whatever the macroblock type, so the compiler can count them and wait for
// "all but the newest six" (s_waitcnt vmcnt(N)) instead of draining the queue.
// Loads a macroblock does not need go to harmless valid addresses (its own
// destination rows, the head of the coefficient buffer) and their results are ignored.
MPG_HD void mb_issue_loads_static(const VideoArgs &a, const MbU &u, int lane, MbLoads &ld)
{
    const int b0 = lane >> 3, j = lane & 7;
    const int b = b0 < 6 ? b0 : 5;
    const bool intra = (u.flags & MPEGHIP_MB_INTRA) != 0;
    int32_t mvx = intra ? 0 : u.mv_x, mvy = intra ? 0 : u.mv_y;
    int32_t stride, off;
    if (b < 4) {
        stride = (int32_t)a.luma_w;
        const int32_t y = (int32_t)(u.mb_y << 4) + j + ((b >> 1) << 3) + (mvy >> 1);
        const int32_t x = (int32_t)(u.mb_x << 4) + ((b & 1) << 3) + (mvx >> 1);
        off = y * stride + x;
    } else {
        mvx /= 2;
        mvy /= 2;
        stride = (int32_t)a.chroma_w;
        const int32_t y = (int32_t)(u.mb_y << 3) + j + (mvy >> 1);
        const int32_t x = (int32_t)(u.mb_x << 3) + (mvx >> 1);
        off = (int32_t)(a.luma_bytes + (b == 5 ? a.chroma_bytes : 0)) + y * stride + x;
    }
    const uint8_t *src = (intra ? (const uint8_t *)u.cur : u.ref) + off;
    const int32_t dy = (mvy & 1) ? stride : 0;
    ld.r0 = ld128u(src);
    ld.r1 = ld128u(src + dy);

    const bool coded = b0 < 6 && (u.cbp & (0x20u >> b)) != 0;
    const bool raw = (u.flags & MPEGHIP_MB_COEF_RAW) != 0;
    const uint32_t k = popc6(u.cbp >> (6 - b));
    const uint8_t *cp = a.coefs + ((uint64_t)u.coef_off + (raw ? 2 * k : k)) * MPEGHIP_COEF_UNIT + (uint32_t)j * (raw ? 32u : 16u);
    if (!coded)
        cp = a.coefs + (uint32_t)lane * 16;
    const uint8_t *cp2 = (coded && raw) ? cp + 16 : u.qm + j * 16;
    ld.c0 = *reinterpret_cast<const i32x4 *>(cp);
    ld.c1 = *reinterpret_cast<const i32x4 *>(cp2);
    ld.qm = ld.pm = 0;
}

// compute half for loads issued by mb_issue_loads_static
MPG_HD void mb_phase_a_compute_static(const VideoArgs &a, const MbU &u, int lane, MbLoads &ld, MbLane &st, int32_t *tile)
{
    if (!(u.flags & MPEGHIP_MB_COEF_RAW)) {
        ld.qm = (uint64_t)(uint32_t)ld.c1.v[0] | ((uint64_t)(uint32_t)ld.c1.v[1] << 32);
        ld.pm = (uint64_t)(uint32_t)ld.c1.v[2] | ((uint64_t)(uint32_t)ld.c1.v[3] << 32);
    }
    mb_phase_a_compute(a, u, lane, ld, st, tile);
}

MPG_HD void mb_phase_a(const VideoArgs &a, const MbU &u, int lane, MbLane &st, int32_t *tile)
{
    MbLoads ld;
    mb_issue_loads(a, u, lane, ld);
    mb_phase_a_compute(a, u, lane, ld, st, tile);
}

// ------------------------------------------------------------------ phase B
// Returns the lane's 8 output bytes (also when nothing is stored) for phase C.
// kStaticStore: every lane (also idle ones) issues exactly one 8-byte store; lanes with
// nothing to write aim it at their private slot of the dump buffer (`sink`).
template <bool kStaticStore>
MPG_HD uint64_t mb_phase_b_t(const VideoArgs &a, const MbU &u, int lane, const MbLane &st, const int32_t *tile, bool &wrote, uint8_t *sink)
{
    wrote = false;
    const int b = lane >> 3, j = lane & 7;
    if (b >= 6) {
        if (kStaticStore)
            *reinterpret_cast<uint64_t *>(sink) = 0;
        return 0;
    }
    const bool intra = (u.flags & MPEGHIP_MB_INTRA) != 0;
    const bool coded = (u.cbp & (0x20u >> b)) != 0;
    if (intra && !coded) {
        if (kStaticStore)
            *reinterpret_cast<uint64_t *>(sink) = 0;
        return 0; // an invalid intra block leaves the old pixels (video.go:711-714)
    }

    uint64_t out = st.pred;
    if (coded) {
        int32_t v[8];
        const i32x4 *t = reinterpret_cast<const i32x4 *>(tile + b * kTileStride + j * 8);
        const i32x4 t0 = t[0], t1 = t[1];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            v[c] = t0.v[c];
            v[c + 4] = t1.v[c];
        }
        idct8<true>(v);
        out = add_clamp_pack8(st.pred, v);
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
    *reinterpret_cast<uint64_t *>(u.cur + off) = out; // 8-byte aligned: x is a multiple of 8
    wrote = true;
    return out;
}

MPG_HD uint64_t mb_phase_b(const VideoArgs &a, const MbU &u, int lane, const MbLane &st, const int32_t *tile, bool &wrote)
{
    return mb_phase_b_t<false>(a, u, lane, st, tile, wrote, nullptr);
}

// ------------------------------------------------------------------ colour
// Go image/draw -> imageutil.DrawYCbCr (4:2:0), as reached from Frame.RGBA
// (video.go:31-36).  Returns R | G<<8 | B<<16 | 255<<24.
MPG_HD uint32_t ycbcr_to_rgba(uint32_t y, uint32_t cb, uint32_t cr)
{
    const int32_t yy1 = (int32_t)y * 0x10101;
    const int32_t cb1 = (int32_t)cb - 128;
    const int32_t cr1 = (int32_t)cr - 128;
    int32_t r = yy1 + 91881 * cr1;
    int32_t g = yy1 - 22554 * cb1 - 46802 * cr1;
    int32_t bl = yy1 + 116130 * cb1;
    r = ((uint32_t)r & 0xff000000u) ? ~(r >> 31) : (r >> 16);
    g = ((uint32_t)g & 0xff000000u) ? ~(g >> 31) : (g >> 16);
    bl = ((uint32_t)bl & 0xff000000u) ? ~(bl >> 31) : (bl >> 16);
    return ((uint32_t)r & 0xff) | (((uint32_t)g & 0xff) << 8) | (((uint32_t)bl & 0xff) << 16) | 0xff000000u;
}

// The same arithmetic arranged for the machine.  The chroma part of r, g, b is shared by the 2 (or 2x2)
// pixels of a chroma sample; "(v & 0xff000000) ? ~(v >> 31) : v >> 16" is clamp(v >> 16, 0, 255), and
// v >> 16 always fits int16 (|v| < 2^25), so the upper halves of r, g (and b, 255) go through
// v_perm_b32 into int16 pairs and v_sat_pk_u8_i16 clamps two channels at once; y * 0x10101 is the
// y byte replicated three times (one v_perm_b32, no extraction).  About 40 clocks per pixel instead of
// about 110 (tools/microbench/valu_rate.hip for the per-instruction costs).
struct ChromaTerms { int32_t r, g, b; };
MPG_HD ChromaTerms chroma_terms(uint32_t cb, uint32_t cr)
{
    const int32_t cb1 = (int32_t)cb - 128, cr1 = (int32_t)cr - 128;
    return ChromaTerms{91881 * cr1, -22554 * cb1 - 46802 * cr1, 116130 * cb1};
}
// pixel k (0..3) of the luma word `yword`
template <int K> MPG_HD uint32_t rgba_pixel(uint32_t yword, const ChromaTerms &c)
{
#if MPG_ON_DEVICE
    const int32_t yy = (int32_t)__builtin_amdgcn_perm(0u, yword, 0x0c000000u | (K << 16) | (K << 8) | K);
    const uint32_t r = (uint32_t)(yy + c.r), g = (uint32_t)(yy + c.g), b = (uint32_t)(yy + c.b);
    const uint32_t rg16 = __builtin_amdgcn_perm(g, r, 0x07060302u); // {r >> 16, g >> 16} as int16
    const uint32_t ba16 = __builtin_amdgcn_perm(0u, b, 0x0c0d0302u); // {b >> 16, 255}
    uint32_t rg8, ba8;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(rg8) : "v"(rg16));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(ba8) : "v"(ba16));
    return __builtin_amdgcn_perm(ba8, rg8, 0x05040100u);             // R, G, B, A in memory order
#else
    const int32_t yy = (int32_t)((yword >> (8 * K)) & 0xff) * 0x10101;
    const int32_t v[3] = {yy + c.r, yy + c.g, yy + c.b};
    uint32_t out = 0xff000000u;
    for (int i = 0; i < 3; i++) {
        const int32_t q = v[i] >> 16; // fits int16
        out |= (uint32_t)(q < 0 ? 0 : (q > 255 ? 255 : q)) << (8 * i);
    }
    return out;
#endif
}
// 4 pixels of one row: luma word + the two chroma samples under it (cb2 / cr2: two bytes each)
MPG_HD void rgba_row4(uint32_t yword, const ChromaTerms &c01, const ChromaTerms &c23, uint32_t (&px)[4])
{
    px[0] = rgba_pixel<0>(yword, c01);
    px[1] = rgba_pixel<1>(yword, c01);
    px[2] = rgba_pixel<2>(yword, c23);
    px[3] = rgba_pixel<3>(yword, c23);
}

// phase C part 1: every lane that holds macroblock pixels parks them in LDS.
// For blocks this macroblock did not write (invalid intra blocks) the current
// frame content is used instead, so the RGBA image always mirrors the planes.
MPG_HD void mb_phase_c_stage(const VideoArgs &a, const MbU &u, int lane, uint64_t out, bool wrote, uint8_t *stage)
{
    const int b = lane >> 3, j = lane & 7;
    if (b >= 6)
        return;
    if (!wrote) {
        uint32_t off;
        if (b < 4)
            off = ((u.mb_y << 4) + (uint32_t)j + ((uint32_t)(b >> 1) << 3)) * a.luma_w + (u.mb_x << 4) + ((uint32_t)(b & 1) << 3);
        else
            off = a.luma_bytes + (b == 5 ? a.chroma_bytes : 0) + ((u.mb_y << 3) + (uint32_t)j) * a.chroma_w + (u.mb_x << 3);
        out = *reinterpret_cast<const uint64_t *>(u.cur + off);
    }
    // stage layout: Y 16 rows x 16 bytes, then Cb 8x8, then Cr 8x8
    uint32_t so;
    if (b < 4)
        so = ((uint32_t)j + ((uint32_t)(b >> 1) << 3)) * 16 + ((uint32_t)(b & 1) << 3);
    else
        so = 256 + (uint32_t)(b - 4) * 64 + (uint32_t)j * 8;
    *reinterpret_cast<uint64_t *>(stage + so) = out;
}

// phase C part 2: lanes 0..31 convert 8 pixels each: lane = row*2 + half.
MPG_HD void mb_phase_c_convert(const VideoArgs &a, const MbU &u, int lane, const uint8_t *stage)
{
    if (lane >= 32)
        return;
    const uint32_t row = (uint32_t)lane >> 1, half = (uint32_t)lane & 1;
    const uint32_t py = (u.mb_y << 4) + row;
    const uint32_t px0 = (u.mb_x << 4) + half * 8;
    if (py >= a.height || px0 >= a.width)
        return;
    const uint64_t yy = *reinterpret_cast<const uint64_t *>(stage + row * 16 + half * 8);
    const uint32_t cb = *reinterpret_cast<const uint32_t *>(stage + 256 + (row >> 1) * 8 + half * 4);
    const uint32_t cr = *reinterpret_cast<const uint32_t *>(stage + 320 + (row >> 1) * 8 + half * 4);
    uint32_t px[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
        px[k] = ycbcr_to_rgba((uint32_t)(yy >> (8 * k)) & 0xff, (cb >> (8 * (k >> 1))) & 0xff, (cr >> (8 * (k >> 1))) & 0xff);
    uint8_t *dst = u.rgba + ((uint64_t)py * a.width + px0) * 4;
    const uint32_t n = a.width - px0 >= 8 ? 8 : a.width - px0;
    if (n == 8 && (((uint64_t)py * a.width + px0) & 3) == 0) {
        u32x4 lo = {{px[0], px[1], px[2], px[3]}}, hi = {{px[4], px[5], px[6], px[7]}};
        reinterpret_cast<u32x4 *>(dst)[0] = lo;
        reinterpret_cast<u32x4 *>(dst)[1] = hi;
    } else {
        for (uint32_t k = 0; k < n; k++)
            reinterpret_cast<uint32_t *>(dst)[k] = px[k];
    }
}

// ------------------------------------------------------- stand-alone Frame.RGBA
// One thread converts a 4x2 block of pixels of one stream's slot (rows 2*yp and 2*yp+1 share their
// chroma samples: one Cb and one Cr word serve 8 pixels).  Grid rows are therefore row PAIRS.
// kStream: non-temporal store — right when a wave writes whole cache lines of an image nobody reads
// on the device (the stand-alone kernel), wrong when a line is completed by later instructions (the
// fused path writes a row in 64-byte pieces and wants L2 to combine them).
template <bool kStream = true>
MPG_HD void rgba_store4(uint32_t *dst, uint64_t p, const uint32_t (&px)[4], uint32_t n)
{
    if (n == 4 && (p & 3) == 0) {
#if MPG_ON_DEVICE
        typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
        const u32v4 q = {px[0], px[1], px[2], px[3]};
        if (kStream)
            __builtin_nontemporal_store(q, reinterpret_cast<u32v4 *>(dst));
        else
            *reinterpret_cast<u32v4 *>(dst) = q;
#else
        u32x4 q = {{px[0], px[1], px[2], px[3]}};
        *reinterpret_cast<u32x4 *>(dst) = q;
#endif
    } else {
        for (uint32_t k = 0; k < n; k++)
            dst[k] = px[k];
    }
}

MPG_HD void rgba_convert_quad(const uint8_t *frame, uint32_t luma_w, uint32_t chroma_w,
                              uint32_t luma_bytes, uint32_t chroma_bytes,
                              uint32_t width, uint32_t height, uint32_t x4, uint32_t yp, uint8_t *rgba)
{
    const uint32_t x0 = x4 * 4, y = yp * 2;
    if (y >= height || x0 >= width)
        return;
    const bool two = y + 1 < height;
    const uint8_t *yrow = frame + (uint64_t)y * luma_w + x0;
    const uint32_t yy0 = *reinterpret_cast<const uint32_t *>(yrow);
    const uint32_t yy1 = *reinterpret_cast<const uint32_t *>(yrow + (two ? luma_w : 0));
    const uint8_t *cbp = frame + luma_bytes + (uint64_t)yp * chroma_w + (x0 >> 1);
    const uint32_t cb = *reinterpret_cast<const uint16_t *>(cbp);
    const uint32_t cr = *reinterpret_cast<const uint16_t *>(cbp + chroma_bytes);
    uint32_t px0[4], px1[4];
    const ChromaTerms c01 = chroma_terms(cb & 0xff, cr & 0xff), c23 = chroma_terms((cb >> 8) & 0xff, (cr >> 8) & 0xff);
    rgba_row4(yy0, c01, c23, px0);
    rgba_row4(yy1, c01, c23, px1);
    const uint64_t p = (uint64_t)y * width + x0;
    uint32_t *dst = reinterpret_cast<uint32_t *>(rgba) + p;
    const uint32_t n = width - x0 >= 4 ? 4 : width - x0;
    rgba_store4(dst, p, px0, n);
    if (two)
        rgba_store4(dst + width, p + width, px1, n);
}

} // namespace mpg
