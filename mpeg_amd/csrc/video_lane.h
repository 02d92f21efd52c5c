// video_lane.h — arithmetic shared by the video kernels: the reference's 8-point IDCT pass, the
// dequantiser, and Frame.RGBA's colour conversion.  The reconstruction kernel itself (lane map,
// device format, host-side packer) is video_recon_lane.h.
//
// Replaces, per macroblock: predictMacroblock/copyMacroblock (video.go:608-637,
// video_noasm.go:28-80, video_amd64.s, video_arm64.s), the dequantise + premultiply tail of
// decodeBlock (video.go:719-744), idct (video.go:801-928) and copy/add*ToDest
// (video.go:943-1002); optionally Frame.RGBA (video.go:31-36).
#pragma once

#include "lane_common.h"
#include "mpeghip.h"

namespace mpg {

struct VideoArgs {
    uint8_t *frames;              // base of the frame store
    uint8_t *frames_b;            // ... minus kRcDmaBias (video_recon_lane.h): what recon_kernel's waves add their stream's offset to
    uint64_t frame_stride;        // bytes between consecutive (stream, slot) frames
    uint32_t mb_w, mb_h;          // macroblocks per row / column
    uint32_t luma_w, luma_h;      // padded plane sizes = 16 mb_w, 16 mb_h
    uint32_t chroma_w, chroma_h;
    uint32_t luma_bytes, chroma_bytes;
    const mpeghip_pic_desc *pics; // (whole-frame RGBA pass only)
    const uint32_t *chunks;       // kRcChunkDwords per chunk of 4 macroblocks (video_recon_lane.h)
    const uint32_t *words;        // per chunk: one word per coded block, then one per non-zero coefficient
    const uint8_t *qmat;          // [n_streams][64 positions][2 classes]{matrix entry, premultiplier}
    uint32_t n_chunks;
    uint32_t width, height;       // display size (RGBA image)
    uint8_t *rgba;                // base of RGBA images, same (stream, slot) indexing
    uint64_t rgba_stride;
};

// ---- frame layout in HBM.  A slot holds the reference's `base` slice (video.go:338-355: Y | Cb | Cr | pad), but the
// planes are TILED in macroblock order: luma as 16x16 tiles of 256 bytes (row-major inside, 16 bytes per row); the two
// chroma planes as one array of 128-byte PAIRS, macroblock i's 8x8 Cb block (64 bytes, 8 per row) followed by its Cr
// block; the pad (luma_w * 16 zero bytes) and the slack stay as they are.
// Why: motion compensation reads a 17 x 17 (9 x 9) window at an arbitrary position per macroblock.  In a row-major
// plane that is one cache line PER ROW (35 rows -> ~39 lines, 11 useful bytes per line requested), and the number of
// lines a wave requests is what bounded the kernel (profiles/r2t_ab_tiled_access_pattern.txt: the same kernel with a
// tile-like access pattern ran 30 % faster).  Tiled, a window touches 2 x 3 luma lines and ~4 chroma lines (one line
// holds both planes of a macroblock; with the planes apart it was ~6: profiles/r5_ab_*, r6_ab_*: +0.3 .. 1.9 % on every
// leg); a macroblock's output is 256 + 128 CONTIGUOUS bytes.  The linear view of the reference (plane reads, hashes,
// Frame.RGBA, and its reads past a plane's edge) is kept exactly: see linear_to_tiled and the kernel's slow path.
constexpr uint32_t kChromaBlockStep = 128; // bytes from one macroblock's Cb block to the next one's
constexpr uint32_t kChromaCrAt = 64;       // a macroblock's Cr block, behind its Cb block
MPG_HD uint32_t tiled_luma(uint32_t mb_w, uint32_t x, uint32_t y) { return ((y >> 4) * mb_w + (x >> 4)) * 256 + (y & 15) * 16 + (x & 15); }
MPG_HD uint32_t tiled_chroma(uint32_t mb_w, uint32_t x, uint32_t y) { return ((y >> 3) * mb_w + (x >> 3)) * kChromaBlockStep + (y & 7) * 8 + (x & 7); }
// byte offset inside a slot in the reference's (linear) layout -> where that byte lives.  A dword-aligned linear dword
// stays one dword (tile rows are 16 / 8 bytes and plane widths multiples of them).
MPG_HD uint32_t linear_to_tiled(uint32_t mb_w, uint32_t luma_bytes, uint32_t chroma_bytes, uint32_t L)
{
    const uint32_t luma_w = mb_w * 16, chroma_w = mb_w * 8;
    if (L < luma_bytes)
        return tiled_luma(mb_w, L % luma_w, L / luma_w);
    uint32_t c = L - luma_bytes, plane = luma_bytes;
    if (c >= chroma_bytes) {
        c -= chroma_bytes;
        plane += kChromaCrAt;
        if (c >= chroma_bytes)
            return L; // pad / slack: linear
    }
    return plane + tiled_chroma(mb_w, c % chroma_w, c / chroma_w);
}

// Descriptors are read-only for the whole launch.  On the device they are read through the constant
// address space so that the compiler keeps them as scalar (s_load) instructions.
#if MPG_ON_DEVICE
#define MPG_CONST_AS __attribute__((address_space(4)))
#else
#define MPG_CONST_AS
#endif

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
    // (row pass: the final `+ 128 >> 8` of video.go:916-925.  Every one of the eight outputs is m0 plus or minus the other
    // terms — m0 enters each exactly once, with a plus sign — so the 128 is added to m0 once instead of to every output:
    // the same integers, seven additions fewer per pass.  tests/test_int_ranges.py carries the bound.)
    const int32_t m0 = kFinalShift ? v[0] + 128 : v[0];
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
            v[k] >>= 8;
    }
}

// Dequantise one quantised level the way the VLC loop does (video.go:719-741): the clamped level, |.| <= 2048.  q != 0.
// qsqm = quantiser_scale * matrix entry.  (The premultiplication of video.go:744 happens when a column is read from the int16
// tile: rc_cols_load16.)
MPG_HD int32_t dequant_level(int32_t q, bool intra, int32_t qsqm)
{
    // level = 2q (+ sign(q) unless intra); q != 0 so sign(q) = (q >> 31) | 1
    int32_t l = 2 * q;
    if (!intra)
        l += (q >> 31) | 1;
    l = mul24_as_written(l, qsqm) >> 4; // |l| <= 65535, qsqm <= 31*255
    // "if even, move one toward zero; 0 becomes +1" == (l - (l > 0)) | 1
    l = (l - (l > 0 ? 1 : 0)) | 1;
    return clampi(l, -2048, 2047);
}

MPG_HD uint32_t popc6(uint32_t x) { return (uint32_t)__builtin_popcount(x & 0x3f); }

// ------------------------------------------------------------------ colour
// Go image/draw -> imageutil.DrawYCbCr (4:2:0), as reached from Frame.RGBA
// (video.go:31-36).  Returns R | G<<8 | B<<16 | 255<<24.  (The reference's form; the device uses the
// arrangement below, proven equal for all 2^24 inputs.)
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

MPG_HD void rgba_convert_quad(const uint8_t *frame, uint32_t mb_w, uint32_t luma_bytes, uint32_t chroma_bytes,
                              uint32_t width, uint32_t height, uint32_t x4, uint32_t yp, uint8_t *rgba)
{
    (void)chroma_bytes; // (the planes' blocks are interleaved per macroblock: no plane stride)
    const uint32_t x0 = x4 * 4, y = yp * 2;
    if (y >= height || x0 >= width)
        return;
    const bool two = y + 1 < height;
    // (a quad of 4 luma pixels lies inside one tile row, its 2 chroma samples inside one block row)
    const uint32_t yy0 = *reinterpret_cast<const uint32_t *>(frame + tiled_luma(mb_w, x0, y));
    const uint32_t yy1 = *reinterpret_cast<const uint32_t *>(frame + tiled_luma(mb_w, x0, two ? y + 1 : y));
    const uint8_t *cbp = frame + luma_bytes + tiled_chroma(mb_w, x0 >> 1, yp);
    const uint32_t cb = *reinterpret_cast<const uint16_t *>(cbp);
    const uint32_t cr = *reinterpret_cast<const uint16_t *>(cbp + kChromaCrAt);
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
