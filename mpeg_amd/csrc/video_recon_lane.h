// video_recon_lane.h — the reconstruction kernel: device format, host-side packer, lane functions.
//
// One WAVE reconstructs a CHUNK = 4 consecutive macroblocks of one picture (normally 4 horizontal
// neighbours) — one chunk per wave (mpeghip.hip) — with wave-private LDS and no barrier.  What the kernel reads is not the C ABI's arrays but
// the library's own device format, which the host half of the library (rc_pack_picture, called from the
// validation pass of every submit / upload) writes straight into the buffer the H2D copy reads:
//
//   chunks  32 dwords = ONE 128-byte line per chunk (round 5; 24 dwords until then): 8 header dwords + 4 records of 6 — one round
//           of scalar loads, and everything a wave derives from them is either a scalar the packer has worked out or one vector
//           instruction away (the round-4 form cost a wave 88 scalar instructions before its first load and ~45 per macroblock of
//           motion compensation: DESIGN.md section 5.0)
//           header  h0 h1  byte offset (64 bits) of the STREAM's first slot in the frame store: the wave's one frame base, for the
//                          four windows and the stores alike
//                   h2 h3  byte offset (64 bits) of the chunk's first block word in the words array (its entries follow its block words)
//                   h4     entries of pass 0 | pass 1 << 10 | pass 2 << 20 | kCRun | kCRgba
//                   h5     coded blocks (0..24) | live macroblocks << 5 | any snapshot block << 8 | any dense block << 9
//                          | any block with its DC in its block word << 10 | destination slot << 11 | stream << 13
//                   h6 h7  a run's destination: byte offsets, from the stream's base, of its four luma tiles / its four Cb | Cr pairs
//                          (+ kRcDmaBias: the scalar base every access of the wave goes through sits that far below the stream)
//           record  r0 kR* flags | cbp << 8 | mb_x << 16 | mb_y << 24
//                   r1 the luma prediction window, origin (x0, y0) = macroblock origin + integer vector: byte offset, from the
//                      stream's base, of the 16-byte piece that holds the window's first row in the 16x16 TILE that holds (x0, y0):
//                      reference slot * frame_stride + tile * 256 + (y0 & 15) * 16
//                   r2 the same for Cb MINUS r1: ... + luma_bytes + pair * 128 + ((cy0 & 7) >> 1) * 16 (its Cr block: 64 further on)
//                   r3 8 (x0 & 3) | (8 (x0 & 3) + 8) << 8 | (x0 & 12) << 16: the luma taps' funnel shifts and the dword they start in
//                   r4 8 (cx0 & 3) | (8 (cx0 & 3) + 8) << 8 | (cy0 & 1) << 16: the chroma taps' shifts, and whether the window starts
//                      on the odd row of its first row pair
//                   r5 (cx0 >> 2) & 1: the dword of a chroma row the window starts in
//                      — every field where the instruction that consumes it looks: a shift amount in the low bits of a dword
//                      (v_alignbit_b32 / v_lshrrev_b64 read 5 / 6 bits), a multiplicand in a 16-bit half (v_mad_u32_u16 op_sel)
//                   kRSlow records (a window that leaves its plane: the reference reads on, linearly, into the next
//                   row / plane / the pad, video_noasm.go:48-80): r1 = the reference slot's offset, r2 = 0, r3 / r4 = the window
//                   origins of the reference's LINEAR layout as column | row << 16 (luma: rows of luma_w bytes from the slot's
//                   start; chroma: Cb | Cr | pad as one array of chroma_w-byte rows from the Cb plane's start), r5 = mb_h;
//                   the kernel gathers those windows dword by dword
//   words   per chunk, one after the other (a wave's loads share cache lines):
//           block words, one per coded block, in (macroblock, block) order = "slot" order:
//                   LDS byte offset / 8 of the block's row 0 in the output bytes | chroma << 9 | snapshot << 10
//                   | dense << 11 | (snapshot / dense: dword offset of its data behind the chunk's first entry) << 12
//                   | (dense: quantiser_scale << 26 | non-intra << 31)
//                   | (sparse intra blocks: the DC level << 12 | kBDcWord — video.go:656-672 treats it apart: no matrix, `<< 8`)
//           entries, one per NON-ZERO quantised coefficient of the sparse blocks, grouped by pass (slots 0-7,
//           8-15, 16-23):
//                   level << 16 | quantiser_scale << 11 | (slot & 7) << 8 | position << 2 | non-intra << 1
//                   ; position = column * 8 + row, the order of the ABI's coefficient units
//           then the data of the chunk's other blocks: int32 snapshot blocks (MPEGHIP_MB_COEF_RAW), 64 dwords
//           each, and DENSE blocks — more than 32 non-zero levels, where a unit as the ABI hands it over
//           (64 int16 levels, 32 dwords) is the shorter form
//
// The reference's VLC loop produces exactly such (position, level) pairs (video.go:680-745); the ABI takes them
// as they are (MPEGHIP_PIC_SPARSE: a pair is an entry short of three bit fields) or as dense 128-byte units, of
// which the packer drops the zeros again.  Dequantisation, premultiply, IDCT, prediction and write-back all
// happen on the device:
//
//   1  scalar loads: header + 4 records.  Then SEVEN vector loads per wave, all issued before the first use: the
//      first 64 entries and the first pass's block words (one dword per lane each, into registers), and five
//      direct-to-LDS loads (global_load_lds_dwordx4: lane l's 16 bytes land at LDS base + 16 l, no registers in
//      between, scalar base + 32-bit lane offset): the stream's dequantisation table (12 lanes) and per macroblock its
//      whole prediction window as 54 PIECES: 17 luma rows x 2 tile rows (16 bytes each) + per chroma plane 5 row
//      PAIRS x 2 blocks (a 16-byte piece = 2 rows of an 8x8 block).  Pieces are whole tile rows; the window's byte
//      offset inside them (wave-uniform) is applied when the taps are read.  No registers hold prediction data (the first version of this kernel kept 32 of them and needed
//      19 load instructions per wave, with ds_bpermute for the row below).
//   2  residual pass (8 coded blocks at a time): zero the wave's tile T[8][64]; one entry per lane:
//      dequantise (video.go:719-744) and scatter to T[slot & 7][position]; an intra block's DC from its block word; lane
//      (g, j): column j of block g, column pass, transposition, row j, row pass (+128 >> 8).  Dense units are dequantised
//      straight from the words, two levels at a time on packed 16-bit halves.  The tile is int16 in both kernel instances; they
//      differ in the transposition between the two IDCT passes (below: across lanes by DPP / through LDS in two halves).
//   3  motion compensation, per macroblock with WAVE-UNIFORM half-pel modes (video_noasm.go:48-80): luma
//      by 64 lanes x 4 pixels, chroma by 32 lanes x 4 pixels, taps read from the window in LDS, result written
//      over it (every lane has its taps before any lane writes): the macroblock's 384 output bytes O_m.
//      Intra macroblocks put zeros.
//   4  lane (g, j) adds its residual row to the 8 prediction bytes in O and clamps (video.go:943-971).
//      (Passes 1, 2 of a chunk with more than 8 coded blocks repeat steps 2 and 4.)
//   5  the four O_m leave as 1 024 contiguous luma / 512 chroma bytes when the chunk is a run (kCRun: 4 macroblocks consecutive
//      in raster order = 4 consecutive tiles, also across a row end), else as 8-byte rows per block; pictures flagged
//      MPEGHIP_PIC_RGBA are colour-converted from them (4 macroblocks wide if the run sits in one row, else per macroblock).
//
// Wave-private LDS, 4 672 or 4 800 bytes (32 waves per CU = 8 per SIMD in both instances):
//      [   0,  192) table    [ 192 + 864 m, + 864) window m -> O_m    [3648, 4672 / 4800) T (int16 tile / transposition buffer over it)
#pragma once

#include "video_lane.h"

#if !MPG_ON_DEVICE && defined(__SSE2__)
#include <emmintrin.h>
#endif
#if !MPG_ON_DEVICE && defined(__x86_64__)
#include <immintrin.h>
#define MPG_HOST_AVX512 1 // compiled in as separate target("avx512...") functions, chosen at run time
#else
#define MPG_HOST_AVX512 0
#endif

namespace mpg {

constexpr int kRcMbs = 4;                     // macroblocks per chunk = per wave
constexpr int kRcMaxBlocks = 6 * kRcMbs;      // 24 slots, 3 passes of 8
constexpr int kRcHeadDwords = 8, kRcRecDwords = 6;
constexpr int kRcChunkDwords = kRcHeadDwords + kRcRecDwords * kRcMbs; // 32: one 128-byte line
constexpr uint32_t kCRun = 1u << 30, kCRgba = 1u << 31;                                    // header h4
// header h5
constexpr uint32_t kHBlocksMask = 0x1f, kHLiveShift = 5, kHAnyRaw = 1u << 8, kHAnyDense = 1u << 9, kHAnyDcWord = 1u << 10, kHCurShift = 11,
                   kHStreamShift = 13;
constexpr uint32_t kRcMaxStreams = 1u << (32 - kHStreamShift); // 524 288 per frame store
// Every access a wave makes to the frame store goes through ONE scalar base, the stream's first slot MINUS this: the four
// window loads carry their LDS target in the instruction's offset field, which the hardware adds to the global address as
// well (lane_common.h: dma16x5_to_lds) — with the base moved back once, by more than the largest of those offsets, the
// correction is a constant in each lane's piece offset instead of a 64-bit scalar subtraction per window.
constexpr uint32_t kRcDmaBias = 4096;
constexpr uint32_t kRIntra = 1, kRDead = 2, kROhL = 4, kROvL = 8, kROhC = 16, kROvC = 32, kRSlow = 64; // record d0
constexpr uint32_t kBChroma = 1u << 9, kBRaw = 1u << 10, kBDense = 1u << 11, kBDcWord = 1u << 28;    // block word
constexpr uint32_t kDenseAbove = 32; // non-zero levels beyond which a block travels as a dense unit
constexpr int32_t kRcDenseLevelMax = 528;  // ... if none of them is beyond this: the dense path's packed 16-bit forms hold 2 * 31 * level + 31
                                             // (rc_dense_cols<true>; an MPEG-1 level is within +-255: video.go:700-707)
constexpr uint32_t kENonIntra = 2;                                                         // entry

// wave-private LDS
constexpr int kRcPiece = 16;                  // bytes per lane of a direct-to-LDS load
constexpr int kRcWinLuma = 17 * 2 * kRcPiece; // 544: 17 rows x 32 bytes
constexpr int kRcWinLanes = 54;               // + 2 planes x 5 row pairs x 2 blocks
constexpr int kRcWinBytes = kRcWinLanes * kRcPiece; // 864
constexpr int kRcQtabBytes = 192;             // [64 positions][{intra, non-intra} matrix entry] + [64] premultiplier
constexpr int kRcQtabStride = 256;            // per stream in HBM
constexpr int kRcQtabAt = 0;
constexpr int kRcWinAt = kRcQtabBytes;        // 192
constexpr int kRcTileAt = kRcWinAt + 4 * kRcWinBytes; // 3648
// The wave's coefficient tile T: int16 [8 blocks][64], 1 024 bytes — dequantised levels (|.| <= 2048) of the pass's SPARSE blocks,
// premultiplied when a column is read (byte x half-word multiplies); snapshot blocks and dense units are read straight from
// HBM.  Two instances of the kernel (mpeghip.hip picks one per batch) differ in how the 8 x 8 transposition between the two
// IDCT passes is done and in what they carry for dense units:
//   kT16 = true   across the block's 8 lanes by DPP (28 vector instructions, no LDS round trip); the general dequantisation of
//                 dense units only.  4 672 bytes of LDS.  The instance for typical batches (a dense unit here and there).
//   kT16 = false  through LDS, in two halves of 4 blocks over the (by then dead) tile: [4][72] int32 — the stride keeps the 32
//                 lanes of a store instruction on 32 banks —, 8 stores + 2 16-byte loads per half, no vector-ALU work; plus the
//                 short dequantisation of dense non-intra units under the default matrix (rc_dense_cols<true>).  4 800 bytes.
//                 The instance for batches with dense units (bound by vector-ALU issue).  (The name is history: rounds 2 - 5
//                 ran this instance on an int32 tile [8][64] of dequantised AND premultiplied values, 5 696 bytes = 7 waves per
//                 SIMD, and transposed through that.  Each resident wave is worth 5 % here — profiles/round5_o_*: 7 -> 6 waves
//                 -5.3 % dense, 8 -> 7 waves -5.7 % typical — and this form runs 8: dense +1.6 %, batches of mixed content
//                 +3 .. 4 %, ahead of BOTH old instances at every share of dense blocks: profiles/round5_p_*.)
// Both run 8 waves per SIMD (32 one-wave workgroups per CU: 160 000 usable bytes, tools/microbench/lds_residency.hip).
constexpr int kRcTileBytes16 = 8 * 64 * 2;
constexpr int kRcTposeStride = 72;                           // dwords per block of the transposition buffer
constexpr int kRcTposeBytes = 4 * kRcTposeStride * 4;        // 1 152
template <bool kT16> constexpr int rc_lds_bytes() { return kRcTileAt + (kT16 ? kRcTileBytes16 : kRcTposeBytes); }
constexpr int kRcLdsBytesMax = kRcTileAt + kRcTposeBytes;

// LDS byte offset of macroblock m's window, later its output bytes O_m: luma [16 rows][16] | Cb [8][8] | Cr [8][8]
MPG_HD uint32_t rc_win_at(uint32_t m) { return kRcWinAt + m * kRcWinBytes; }

// LDS byte offset of row j of block b of macroblock m (inside O_m)
MPG_HD uint32_t rc_tile_offset(int b, int j, uint32_t m)
{
    if (b < 4)
        return rc_win_at(m) + ((uint32_t)j + ((uint32_t)(b >> 1) << 3)) * 16 + ((uint32_t)(b & 1) << 3);
    return rc_win_at(m) + 256 + (uint32_t)(b - 4) * 64 + (uint32_t)j * 8;
}

// ===================================================================== host half: the packer
struct RcGeom {
    uint32_t mb_w, mb_h, luma_w, chroma_w, luma_bytes;
    uint64_t frame_stride, rgba_stride;
};

// One stream's device table (kRcQtabStride bytes, kRcQtabBytes used): [position = col*8+row]{intra, non-intra matrix
// entry}, then [position] premultiplier
static inline void rc_make_qtable(uint8_t out[256], const uint8_t intra[64], const uint8_t non_intra[64], const uint8_t premult[64])
{
    memset(out, 0, kRcQtabStride);
    for (int c = 0; c < 8; c++)
        for (int r = 0; r < 8; r++) {
            out[(c * 8 + r) * 2 + 0] = intra[r * 8 + c];
            out[(c * 8 + r) * 2 + 1] = non_intra[r * 8 + c];
            out[128 + c * 8 + r] = premult[r * 8 + c];
        }
}

// Which of a unit's 64 int16 words are non-zero (bit k <=> word k).
static inline uint64_t rc_nonzero_mask(const uint8_t *unit)
{
#if !MPG_ON_DEVICE && defined(__SSE2__)
    const __m128i zero = _mm_setzero_si128();
    uint64_t mask = 0;
    for (int k = 0; k < 4; k++) { // 16 words per step
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(unit + k * 32));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(unit + k * 32 + 16));
        const __m128i z = _mm_packs_epi16(_mm_cmpeq_epi16(a, zero), _mm_cmpeq_epi16(b, zero)); // 0xff where zero
        mask |= (uint64_t)(uint16_t)~_mm_movemask_epi8(z) << (k * 16);
    }
    return mask;
#else
    uint64_t mask = 0;
    for (uint32_t k = 0; k < 64; k++) {
        uint16_t w;
        memcpy(&w, unit + k * 2, 2);
        mask |= (uint64_t)(w != 0) << k;
    }
    return mask;
#endif
}

// Can a unit travel as a dense unit?  The dense path forms 2 level + sign(level) in 16 bits (rc_dense_pair): every level
// within +-kRcDenseLevelMax; an intra block's DC does not take that route (rc_dense_cols).
static inline bool rc_dense_levels_fit(const uint8_t *unit, bool intra)
{
    int16_t w[64];
    memcpy(w, unit, 128);
    if (intra)
        w[0] = 0;
    int32_t lo = 0, hi = 0;
    for (int k = 0; k < 64; k++) { // (vectorises: two 512-bit min / max)
        lo = w[k] < lo ? w[k] : lo;
        hi = w[k] > hi ? w[k] : hi;
    }
    return lo >= -kRcDenseLevelMax && hi <= kRcDenseLevelMax;
}

// Room one picture of n macroblocks with `units` coefficient units can need (dwords).
#if MPG_HOST_AVX512
// The same two steps on 512-bit registers (run-time dispatch: rc_host_has_avx512).  Entries of one unit: its non-zero
// words as `word << 16 | position << 2 | bits`, ascending positions — sixteen words at a time are widened, tagged with
// their positions and compressed in a register; the store is a plain 64-byte one whose tail the next store (or the
// chunk's next words) overwrites, so `out` needs 15 dwords of slack (rc_max_words budgets 65 per unit, a sparse block
// uses at most 33).
__attribute__((target("avx512f,avx512bw"))) static inline uint64_t rc_nonzero_mask_avx512(const uint8_t *unit)
{
    const __m512i a = _mm512_loadu_si512(unit), b = _mm512_loadu_si512(unit + 64);
    return (uint64_t)_mm512_test_epi16_mask(a, a) | ((uint64_t)_mm512_test_epi16_mask(b, b) << 32);
}
__attribute__((target("avx512f,avx512bw,avx512vl"))) static inline uint32_t rc_emit_entries_avx512(const uint8_t *unit, uint64_t mask,
                                                                                               uint32_t bits, uint32_t *out)
{
    const __m512i pos4 = _mm512_set_epi32(60, 56, 52, 48, 44, 40, 36, 32, 28, 24, 20, 16, 12, 8, 4, 0);
    uint32_t n = 0;
    for (uint32_t g = 0; g < 4; g++) {
        const __mmask16 m = (__mmask16)(mask >> (16 * g)); // (no early-out for an empty group: the branch mispredicts)
        const __m512i w = _mm512_cvtepu16_epi32(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(unit + 32 * g)));
        const __m512i e = _mm512_or_si512(_mm512_slli_epi32(w, 16), _mm512_add_epi32(pos4, _mm512_set1_epi32((int)(bits + 64 * g))));
        _mm512_storeu_si512(out + n, _mm512_maskz_compress_epi32(m, e));
        n += (uint32_t)__builtin_popcount(m);
    }
    return n;
}
static inline bool rc_host_has_avx512()
{
    static const bool have = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    return have;
}
#endif

// ---- one macroblock's record: ONE function for the host packer below and the device packer (video_pack_lane.h), so that the two
// cannot drift apart.  ref_slot: the slot (0..2) the macroblock predicts from (ignored for intra).  The address arithmetic of
// video_noasm.go:28-43 / video.go:747-770 happens here, once per macroblock at pack time, not once per wave at run time.
MPG_HD void rc_make_record(uint32_t mb_w, uint32_t mb_h, uint32_t luma_w, uint32_t chroma_w, uint32_t luma_bytes, uint64_t frame_stride,
                           uint32_t mb_x, uint32_t mb_y, uint32_t cbp, bool intra, int32_t mvx, int32_t mvy, uint32_t ref_slot,
                           uint32_t (&r)[kRcRecDwords])
{
    const uint32_t r0 = (cbp << 8) | (mb_x << 16) | (mb_y << 24);
    r[1] = r[2] = r[3] = r[4] = r[5] = 0;
    if (intra) { // (its window loads fetch the head of the stream's first slot: valid memory, never used)
        r[0] = r0 | kRIntra;
        return;
    }
    const int32_t cmx = mvx / 2, cmy = mvy / 2; // toward zero, video_noasm.go:35-36
    uint32_t f = 0;
    f |= (mvx & 1) ? kROhL : 0;
    f |= (mvy & 1) ? kROvL : 0;
    f |= (cmx & 1) ? kROhC : 0;
    f |= (cmy & 1) ? kROvC : 0;
    // window origins in pixels; inside their planes (the normal case) the tiles are addressed directly
    const int32_t x0 = (int32_t)(mb_x << 4) + (mvx >> 1), y0 = (int32_t)(mb_y << 4) + (mvy >> 1);
    const int32_t cx0 = (int32_t)(mb_x << 3) + (cmx >> 1), cy0 = (int32_t)(mb_y << 3) + (cmy >> 1);
    const bool inside = x0 >= 0 && y0 >= 0 && x0 + 16 + (mvx & 1) <= (int32_t)luma_w && y0 + 16 + (mvy & 1) <= (int32_t)(mb_h << 4) &&
                        cx0 >= 0 && cy0 >= 0 && cx0 + 8 + (cmx & 1) <= (int32_t)chroma_w && cy0 + 8 + (cmy & 1) <= (int32_t)(mb_h << 3);
    const uint32_t ref_off = (uint32_t)(ref_slot * frame_stride); // (3 slots of at most 25 MB: 32 bits hold it)
    if (inside) {
        const uint32_t ux = (uint32_t)x0, uy = (uint32_t)y0, ucx = (uint32_t)cx0, ucy = (uint32_t)cy0;
        const uint32_t xl = ref_off + ((uy >> 4) * mb_w + (ux >> 4)) * 256 + (uy & 15) * 16;
        const uint32_t xc = ref_off + luma_bytes + ((ucy >> 3) * mb_w + (ucx >> 3)) * kChromaBlockStep + ((ucy & 7) >> 1) * 16;
        const uint32_t sl = (ux & 3) * 8, sc = (ucx & 3) * 8;
        r[0] = r0 | f;
        r[1] = xl;
        r[2] = xc - xl;
        r[3] = sl | ((sl + 8) << 8) | ((ux & 12) << 16);
        r[4] = sc | ((sc + 8) << 8) | ((ucy & 1) << 16);
        r[5] = (ucx >> 2) & 1;
    } else { // the reference's linear reads (validated: inside [plane start, end of base))
        const int32_t dst_luma = (int32_t)(mb_y << 4) * (int32_t)luma_w + (int32_t)(mb_x << 4);
        const int32_t dst_chroma = (int32_t)(mb_y << 3) * (int32_t)chroma_w + (int32_t)(mb_x << 3);
        // the window origins as LINEAR byte offsets of the reference's layout (from the slot's start / from the Cb plane's start),
        // handed over as (column, row) of the array they lie in — rows of luma_w bytes from the slot's start; Cb | Cr | pad as ONE
        // array of chroma_w-byte rows — so that the kernel's gather needs no division (round 6: it made two per dword, ~250
        // vector instructions per lane and window; the two here are the packer's, once per macroblock)
        const uint32_t l0 = (uint32_t)(dst_luma + (mvy >> 1) * (int32_t)luma_w + (mvx >> 1));
        const uint32_t c0 = (uint32_t)(dst_chroma + (cmy >> 1) * (int32_t)chroma_w + (cmx >> 1));
        r[0] = r0 | f | kRSlow;
        r[1] = ref_off;
        r[3] = (l0 % luma_w) | ((l0 / luma_w) << 16);
        r[4] = (c0 % chroma_w) | ((c0 / chroma_w) << 16);
        r[5] = mb_h;
    }
}
// the header dwords that do not depend on the chunk's blocks.  mb0: raster index of the chunk's first macroblock (a run's tiles)
MPG_HD void rc_make_header_base(uint64_t frame_stride, uint32_t luma_bytes, uint32_t stream, uint32_t cur_slot, uint32_t mb0, uint32_t (&h)[kRcHeadDwords])
{
    const uint64_t base = (uint64_t)stream * MPEGHIP_SLOTS * frame_stride;
    const uint32_t cur_off = (uint32_t)(cur_slot * frame_stride);
    h[0] = (uint32_t)base;
    h[1] = (uint32_t)(base >> 32);
    h[6] = kRcDmaBias + cur_off + mb0 * 256;
    h[7] = kRcDmaBias + cur_off + luma_bytes + mb0 * kChromaBlockStep;
}
MPG_HD uint32_t rc_header_flags(uint32_t n_slots, uint32_t live, bool any_raw, bool any_dense, bool any_dcword, uint32_t cur_slot, uint32_t stream)
{
    return n_slots | (live << kHLiveShift) | (any_raw ? kHAnyRaw : 0u) | (any_dense ? kHAnyDense : 0u) | (any_dcword ? kHAnyDcWord : 0u) |
           (cur_slot << kHCurShift) | (stream << kHStreamShift);
}
// Is macroblock (x, y) the j-th tile behind the chunk's first macroblock (x0, y0)?  A RUN (kCRun) is 4 macroblocks whose tiles are
// consecutive in the frame store: tiles lie in raster order, so the 4 neighbours of one row are a run — and so is a chunk that
// wraps from the end of one macroblock row to the head of the next (mb_w % 4 != 0: SIF's 22, the golden streams' 10; until round 6
// such a chunk — one in 5.5 at SIF, two in five at 160x120 — took the per-block stores).  The plane stores of a run are 1 024 + 512
// contiguous bytes from h6 / h7 either way; only the fused colour conversion, which writes IMAGE rows, asks whether the run sits
// in one row (rc_run_in_one_row).  One function for both packers.
MPG_HD bool rc_run_follows(uint32_t mb_w, uint32_t x0, uint32_t y0, uint32_t x, uint32_t y, uint32_t j)
{
    return y * mb_w + x == y0 * mb_w + x0 + j;
}
// a chunk that does nothing (a refused device-packed commit; padding)
MPG_HD void rc_make_dead_chunk(uint32_t *h)
{
    for (int i = 0; i < kRcChunkDwords; i++)
        h[i] = 0;
    for (int m = 0; m < kRcMbs; m++)
        h[kRcHeadDwords + kRcRecDwords * m] = kRDead;
}

static inline size_t rc_max_chunks(uint32_t n) { return ((size_t)n + kRcMbs - 1) / kRcMbs; }
static inline size_t rc_max_words(uint64_t units) { return (size_t)units * 65; }
constexpr size_t kRcWordsPad = 256; // dwords behind the last chunk's words that a wave may read (and ignore)
constexpr size_t kRcQtabPad = 1024;  // bytes behind the last stream's table that a wave may read (and ignore)

struct RcPacked {
    uint32_t chunks = 0, words = 0;        // what the picture took
    uint32_t blocks = 0, dense_blocks = 0; // its coded blocks / those that travel as dense units (which kernel instance suits the batch)
    uint32_t bad = 0;                      // sparse input: 1 + the index of the macroblock whose block data is malformed (nothing usable was packed)
};

// ---- the sparse hand-over (mpeghip_video_stage_put_sparse): the coded blocks' data as the reference's VLC loop
// produces it (video.go:680-745) — per sparse block a count word n and n PAIRS `level << 16 | position << 2` (an intra
// block's first pair is its DC, position 0), per snapshot block the count word 64 and its 64 int32 values; mbs[k].coef_off =
// dword index of macroblock k's first word, never before the end of macroblock k - 1's data.  A pair IS a device entry short of the bits the packer adds (quantiser_scale, slot, class).
// The packer checks a picture's words as it walks them (RcPacked::bad): every block inside [0, n_words), macroblocks in order
// and not overlapping, counts <= 64 (= 64 for a snapshot block), no bit outside a pair's two fields, an intra block's DC first.  (A position named twice in one block is not looked for: the
// parser cannot produce it, and the block's result is then merely unspecified — one of the two levels wins.)
// Pairs -> entries: out[i] = pr[i] | bits for i < cnt; returns the OR of the pairs (their stray bits, if any).  The store may
// run up to 15 dwords past out + cnt (the words buffer has the slack); nothing is read past pr + cnt.
#if MPG_HOST_AVX512
__attribute__((target("avx512f,avx512bw,avx512vl"))) static inline uint32_t rc_pairs_to_entries_avx512(const uint32_t *pr, uint32_t cnt,
                                                                                                   uint32_t bits, uint32_t *out)
{
    const __m512i b = _mm512_set1_epi32((int)bits);
    __m512i acc = _mm512_setzero_si512();
    for (uint32_t i = 0; i < cnt; i += 16) {
        const __mmask16 m = cnt - i >= 16 ? (__mmask16)0xffff : (__mmask16)((1u << (cnt - i)) - 1);
        const __m512i w = _mm512_maskz_loadu_epi32(m, pr + i);
        acc = _mm512_or_si512(acc, w);
        _mm512_storeu_si512(out + i, _mm512_or_si512(w, b));
    }
    return (uint32_t)_mm512_reduce_or_epi32(acc);
}
#endif
static inline uint32_t rc_pairs_to_entries(const uint32_t *pr, uint32_t cnt, uint32_t bits, uint32_t *out)
{
    uint32_t acc = 0;
    for (uint32_t i = 0; i < cnt; i++) {
        acc |= pr[i];
        out[i] = pr[i] | bits;
    }
    return acc;
}
// room the packed form of a sparse picture can need (dwords): its input words (an entry per pair, a block word per count
// word, a unit where it is the shorter form: blocks do not share words, so no more than came in) + the slack wide stores
// run into
static inline size_t rc_max_words_sparse(uint64_t n_words, uint32_t n_mbs) { (void)n_mbs; return (size_t)n_words + 64; }

// Pack ONE picture: macroblocks mbs[0..n) (already validated), whose coef_off index 128-byte units behind
// `coefs` (kSparseIn: the n_sparse dwords of the sparse hand-over behind `coefs`, checked on the way: RcPacked::bad).  Chunk headers name
// their words by index: this picture's first word is word_base (callers that only learn the base afterwards pass 0
// and add it with rc_rebase).
template <bool kWide = true, bool kSparseIn = false> // kWide: use the 512-bit forms where the CPU has them (tests compare both)
static inline RcPacked rc_pack_picture(const RcGeom &g, const mpeghip_pic_desc &p, const mpeghip_mb_desc *mbs, uint32_t n,
                                       const uint8_t *coefs, uint32_t word_base, uint32_t *chunks_out, uint32_t *words_out,
                                       uint64_t n_sparse = 0, uint64_t out_room = ~0ull >> 1)
{
    // out_room (kSparseIn): dwords of room behind words_out.  One picture's packed form is never longer than its input (its
    // macroblocks do not share words), but the pictures of one submit may name the same words: the room is checked block by
    // block and a picture that would not fit is refused (RcPacked::bad) instead of overrunning the buffer.
    RcPacked out;
    const uint32_t *sparse = reinterpret_cast<const uint32_t *>(coefs);
    uint64_t sparse_end = 0; // kSparseIn: where the previous macroblock's data ended (dwords)
    (void)sparse;
    (void)n_sparse;
    (void)sparse_end;
#if MPG_HOST_AVX512
    const bool wide = kWide && rc_host_has_avx512();
#endif
    const bool rgba = (p.flags & MPEGHIP_PIC_RGBA) != 0;
    for (uint32_t k0 = 0; k0 < n; k0 += kRcMbs) {
        const uint32_t live = n - k0 < (uint32_t)kRcMbs ? n - k0 : (uint32_t)kRcMbs;
        uint32_t *h = chunks_out + (size_t)out.chunks * kRcChunkDwords;
        // one pass over the chunk's coded blocks: the block words come first, so their number is counted up front
        uint32_t n_coded = 0;
        for (uint32_t m = 0; m < live; m++)
            n_coded += (uint32_t)__builtin_popcount(mbs[k0 + m].cbp & 0x3fu);
        struct Deferred { // snapshot and dense blocks: their data goes behind the chunk's entries
            const uint8_t *unit;
            uint32_t slot, dwords;
        } deferred[kRcMaxBlocks];
        alignas(16) int16_t built[kSparseIn ? kRcMaxBlocks : 1][64]; // kSparseIn: units made of a block's pairs
        uint32_t n_slots = 0, n_deferred = 0, deferred_dwords = 0;
        uint32_t *bw = words_out + out.words;
        if (kSparseIn && (uint64_t)out.words + n_coded + 16 > out_room) {
            out.bad = k0 + 1;
            return out;
        }
        uint32_t *e0 = bw + n_coded, ne = 0, counts = 0, pass_start = 0;
        bool any_raw = false, any_dense = false, any_dcword = false;
        bool run = live == (uint32_t)kRcMbs; // 4 macroblocks consecutive in raster order = 4 consecutive tiles (rc_run_follows)
        for (uint32_t m = 0; m < (uint32_t)kRcMbs; m++) {
            uint32_t *d = h + kRcHeadDwords + m * kRcRecDwords;
            if (m >= live) { // padding behind the picture's last macroblock
                d[0] = kRDead;
                d[1] = d[2] = d[3] = d[4] = d[5] = 0;
                continue;
            }
            const mpeghip_mb_desc &mb = mbs[k0 + m];
            const bool intra = (mb.flags & MPEGHIP_MB_INTRA) != 0, raw = (mb.flags & MPEGHIP_MB_COEF_RAW) != 0;
            {
                uint32_t rec[kRcRecDwords];
                rc_make_record(g.mb_w, g.mb_h, g.luma_w, g.chroma_w, g.luma_bytes, g.frame_stride, mb.mb_x, mb.mb_y, mb.cbp, intra, mb.mv_x,
                               mb.mv_y, (mb.flags & MPEGHIP_MB_REF_BWD) ? p.bwd : p.fwd, rec);
                for (int i = 0; i < kRcRecDwords; i++)
                    d[i] = rec[i];
            }
            run = run && rc_run_follows(g.mb_w, mbs[k0].mb_x, mbs[k0].mb_y, mb.mb_x, mb.mb_y, m) &&
                  (!intra || mb.cbp == 0x3f); // an invalid intra block keeps the old pixels: no whole rows
            uint64_t unit = mb.coef_off;
            if (kSparseIn) { // a macroblock's data begins where the previous one's ended, or later (every macroblock's coef_off
                             // takes part, coded blocks or not): no two blocks share words, so what is packed is never longer
                             // than what came in, and the device-side packer (video_pack_lane.h) places a chunk's words by the
                             // offset of its first macroblock
                if (unit < sparse_end || unit > n_sparse) {
                    out.bad = k0 + m + 1;
                    return out;
                }
            }
            for (uint32_t left = mb.cbp & 0x3fu; left; ) { // coded blocks in block order = from bit 5 down (one exit branch)
                const int b = __builtin_clz(left) - 26;
                left &= ~(0x20u >> b);
                const uint8_t *u = coefs + (size_t)unit * MPEGHIP_COEF_UNIT;
                const uint32_t s = n_slots++;
                if ((s & 7) == 0 && s) { // a new pass of 8 blocks starts: its entries are counted separately
                    counts |= (ne - pass_start) << (10 * ((s >> 3) - 1));
                    pass_start = ne;
                }
                bw[s] = (rc_tile_offset(b, 0, m) >> 3) | (b >= 4 ? kBChroma : 0u) | (raw ? kBRaw : 0u);
                if (kSparseIn) { // (`unit` counts dwords here, and may run to 2^32 + 64 on malformed input: 64 bits)
                    const uint32_t *sp = sparse + unit;
                    if (raw) { // a snapshot block: the count word says 64, then its 64 int32 values
                        if (unit + 65 > n_sparse || sp[0] != 64) {
                            out.bad = k0 + m + 1;
                            return out;
                        }
                        any_raw = true;
                        deferred[n_deferred++] = Deferred{reinterpret_cast<const uint8_t *>(sp + 1), s, 64};
                        deferred_dwords += 64;
                        unit += 65;
                        if ((uint64_t)out.words + n_coded + ne + deferred_dwords + 16 > out_room) {
                            out.bad = k0 + m + 1;
                            return out;
                        }
                        continue;
                    }
                    uint32_t cnt = unit < n_sparse ? sp[0] : 65;
                    const uint32_t *pr = sp + 1;
                    if (cnt > 64 || unit + 1 + cnt > n_sparse || (intra && (cnt == 0 || (pr[0] & 0xfcu))) ||
                        (uint64_t)out.words + n_coded + ne + deferred_dwords + cnt + 16 > out_room) {
                        out.bad = k0 + m + 1;
                        return out;
                    }
                    unit += 1 + cnt;
                    // more than kDenseAbove levels: a unit is the shorter form — if the dense path can take them: every level
                    // non-zero (a coded zero level dequantises to +-1, video.go:719-736: only an entry says that) and
                    // within its 16-bit steps; an intra block's DC is exempt from both
                    bool as_unit = cnt > kDenseAbove;
                    uint32_t stray = 0;
                    for (uint32_t i = 0; as_unit && i < cnt; i++) {
                        const int32_t level = (int16_t)(pr[i] >> 16);
                        stray |= pr[i];
                        as_unit = (intra && i == 0) || (level != 0 && level >= -kRcDenseLevelMax && level <= kRcDenseLevelMax);
                    }
                    if (as_unit) {
                        if (stray & 0xff03u) {
                            out.bad = k0 + m + 1;
                            return out;
                        }
                        int16_t *bu = built[n_deferred];
                        memset(bu, 0, 128);
                        for (uint32_t i = 0; i < cnt; i++)
                            bu[(pr[i] >> 2) & 63] = (int16_t)(pr[i] >> 16);
                        any_dense = true;
                        out.dense_blocks++;
                        bw[s] |= kBDense | ((uint32_t)(mb.qscale & 31) << 26) | (intra ? 0u : 1u << 31);
                        deferred[n_deferred++] = Deferred{reinterpret_cast<const uint8_t *>(bu), s, 32};
                        deferred_dwords += 32;
                        continue;
                    }
                    stray = 0;
                    if (intra) { // the DC pair comes first; it rides in the block word
                        stray = pr[0];
                        bw[s] |= kBDcWord | ((pr[0] >> 16) << 12);
                        any_dcword = true;
                        pr++;
                        cnt--;
                    }
                    const uint32_t bits = ((uint32_t)(mb.qscale & 31) << 11) | (intra ? 0u : kENonIntra) | ((s & 7) << 8);
#if MPG_HOST_AVX512
                    stray |= wide ? rc_pairs_to_entries_avx512(pr, cnt, bits, e0 + ne) : rc_pairs_to_entries(pr, cnt, bits, e0 + ne);
#else
                    stray |= rc_pairs_to_entries(pr, cnt, bits, e0 + ne);
#endif
                    if (stray & 0xff03u) {
                        out.bad = k0 + m + 1;
                        return out;
                    }
                    ne += cnt;
                    continue;
                }
                unit += raw ? 2 : 1;
                if (raw) {
                    any_raw = true;
                    deferred[n_deferred++] = Deferred{u, s, 64};
                    continue;
                }
#if MPG_HOST_AVX512
                uint64_t mask = wide ? rc_nonzero_mask_avx512(u) : rc_nonzero_mask(u);
#else
                uint64_t mask = rc_nonzero_mask(u);
#endif
                // the unit as it is is the shorter form — if the dense path's 16-bit steps can hold its levels (else: entries)
                const bool as_unit = (uint32_t)__builtin_popcountll(mask) > kDenseAbove && rc_dense_levels_fit(u, intra);
                if (intra && !as_unit) { // an intra block's DC rides in its block word (the int16 tile holds AC levels only)
                    uint16_t dc;
                    memcpy(&dc, u, 2);
                    bw[s] |= kBDcWord | ((uint32_t)dc << 12);
                    any_dcword = true;
                    mask &= ~1ull;
                }
                if (as_unit) {
                    any_dense = true;
                    out.dense_blocks++;
                    bw[s] |= kBDense | ((uint32_t)(mb.qscale & 31) << 26) | (intra ? 0u : 1u << 31);
                    deferred[n_deferred++] = Deferred{u, s, 32};
                    continue;
                }
                const uint32_t bits = ((uint32_t)(mb.qscale & 31) << 11) | (intra ? 0u : kENonIntra) | ((s & 7) << 8);
#if MPG_HOST_AVX512
                if (wide) {
                    ne += rc_emit_entries_avx512(u, mask, bits, e0 + ne);
                    continue;
                }
#endif
                for (uint64_t left_bits = mask; left_bits; left_bits &= left_bits - 1) {
                    const uint32_t pos = (uint32_t)__builtin_ctzll(left_bits);
                    uint16_t w;
                    memcpy(&w, u + pos * 2, 2);
                    e0[ne++] = ((uint32_t)w << 16) | bits | (pos << 2);
                }
            }
            if (kSparseIn)
                sparse_end = unit;
        }
        if (n_slots)
            counts |= (ne - pass_start) << (10 * ((n_slots - 1) >> 3));
        for (uint32_t i = 0; i < n_deferred; i++) { // the unit(s) as they are (position order = the unit's order)
            bw[deferred[i].slot] |= ne << 12;
            memcpy(e0 + ne, deferred[i].unit, deferred[i].dwords * 4);
            ne += deferred[i].dwords;
        }
        {
            uint32_t hd[kRcHeadDwords];
            rc_make_header_base(g.frame_stride, g.luma_bytes, p.stream, p.cur, (uint32_t)mbs[k0].mb_y * g.mb_w + mbs[k0].mb_x, hd);
            const uint64_t wat = ((uint64_t)word_base + out.words) * 4;
            h[0] = hd[0], h[1] = hd[1];
            h[2] = (uint32_t)wat, h[3] = (uint32_t)(wat >> 32);
            h[4] = counts | (run ? kCRun : 0u) | (rgba ? kCRgba : 0u);
            h[5] = rc_header_flags(n_slots, live, any_raw, any_dense, any_dcword, p.cur, p.stream);
            h[6] = hd[6], h[7] = hd[7];
        }
        out.chunks++;
        out.words += n_slots + ne;
        out.blocks += n_slots;
    }
    return out;
}

// add the base a picture's chunks were packed without
static inline void rc_rebase(uint32_t *chunks, uint32_t n_chunks, uint32_t word_base)
{
    for (uint32_t c = 0; c < n_chunks; c++) {
        uint32_t *h = chunks + (size_t)c * kRcChunkDwords;
        const uint64_t wat = ((uint64_t)h[2] | ((uint64_t)h[3] << 32)) + (uint64_t)word_base * 4;
        h[2] = (uint32_t)wat, h[3] = (uint32_t)(wat >> 32);
    }
}
// dword index of a chunk's first block word (tests, the emulator's statistics)
static inline uint64_t rc_chunk_word_index(const uint32_t *h) { return ((uint64_t)h[2] | ((uint64_t)h[3] << 32)) / 4; }

// ===================================================================== device half: lane functions
// (also compiled by g++ into tests/kernel_emu, which runs the 64 lanes of a wave in a loop, phase by phase)

struct RcChunk {
    uint64_t off[2];           // h0 h1, h2 h3 as the two 64-bit offsets they are (one scalar add + add-with-carry each)
    uint32_t h[kRcHeadDwords]; // (h[4..7]; 0..3 are not kept)
    uint32_t r[kRcMbs][kRcRecDwords];
};

MPG_HD RcChunk rc_load_chunk(const VideoArgs &a, uint32_t chunk)
{
    // (chunk * 128 bytes: a 32-bit product — launch_batch refuses batches of 2^25 chunks and more)
    const MPG_CONST_AS uint32_t *p = (const MPG_CONST_AS uint32_t *)((uintptr_t)a.chunks + (uint32_t)(chunk * (kRcChunkDwords * 4)));
    RcChunk c;
    c.off[0] = *reinterpret_cast<const MPG_CONST_AS uint64_t *>(p);
    c.off[1] = *reinterpret_cast<const MPG_CONST_AS uint64_t *>(p + 2);
#pragma unroll
    for (int i = 0; i < 4; i++)
        c.h[i] = 0;
#pragma unroll
    for (int i = 4; i < kRcHeadDwords; i++)
        c.h[i] = p[i];
#pragma unroll
    for (int m = 0; m < kRcMbs; m++)
#pragma unroll
        for (int i = 0; i < kRcRecDwords; i++)
            c.r[m][i] = p[kRcHeadDwords + m * kRcRecDwords + i];
    return c;
}

MPG_HD uint32_t rc_n_blocks(const RcChunk &c) { return c.h[5] & kHBlocksMask; }
MPG_HD uint32_t rc_n_live(const RcChunk &c) { return (c.h[5] >> kHLiveShift) & 7; }
MPG_HD bool rc_any_raw(const RcChunk &c) { return (c.h[5] & kHAnyRaw) != 0; }
MPG_HD bool rc_any_dense(const RcChunk &c) { return (c.h[5] & kHAnyDense) != 0; }
MPG_HD bool rc_any_dcword(const RcChunk &c) { return (c.h[5] & kHAnyDcWord) != 0; } // (intra macroblocks only: most chunks skip the test per lane)
MPG_HD bool rc_any_special(const RcChunk &c) { return (c.h[5] & (kHAnyRaw | kHAnyDense | kHAnyDcWord)) != 0; }
MPG_HD uint32_t rc_cur_slot(const RcChunk &c) { return (c.h[5] >> kHCurShift) & 3; }
MPG_HD uint32_t rc_stream(const RcChunk &c) { return c.h[5] >> kHStreamShift; }
// a run whose 4 macroblocks share a macroblock row (mb_y: the top byte of a record's first dword; a run's macroblocks are consecutive
// in raster order, so first and last in one row = all in one row): the fused colour conversion may write its image rows 4
// macroblocks wide.  A run that wraps a row end converts macroblock by macroblock.
MPG_HD bool rc_run_in_one_row(const RcChunk &c) { return ((c.r[0][0] ^ c.r[kRcMbs - 1][0]) >> 24) == 0; }
MPG_HD uint32_t rc_pass_entries(const RcChunk &c, uint32_t pass) { return (c.h[4] >> (10 * pass)) & 0x3ff; }
// the wave's scalar bases: the stream's frames (moved back by kRcDmaBias: every offset below carries that bias) and the chunk's words
MPG_HD uint8_t *rc_frame_base(const VideoArgs &a, const RcChunk &c) { return a.frames_b + c.off[0]; }
MPG_HD const uint32_t *rc_word_base(const VideoArgs &a, const RcChunk &c)
{
    return reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(a.words) + c.off[1]);
}
// the destination slot's offset from that base (chunks that are not runs, the fused colour conversion)
// (h6 = that + 256 * the raster index of the chunk's first macroblock: no stride, no multiplication by the slot — the kernel keeps
// nothing of the geometry in registers but mb_w and luma_bytes)
MPG_HD uint32_t rc_mb_index(const VideoArgs &a, uint32_t d0) { return (d0 >> 24) * a.mb_w + ((d0 >> 16) & 0xff); }
MPG_HD uint32_t rc_cur_offset(const VideoArgs &a, const RcChunk &c) { return c.h[6] - rc_mb_index(a, c.r[0][0]) * 256; }

// ---- 16-bit multiply-adds whose wave-uniform multiplicand sits in one half of a record dword (the packer puts it there): the
// instruction's operand select picks the half, so a field of a record reaches a lane's address with no scalar instruction
// (no s_bfe / s_lshr / s_and in front of it).  kHi: the multiplicand is the dword's upper half.
template <bool kHi> MPG_HD uint32_t mad_u16_field(uint32_t field_dword, uint32_t lane_factor, uint32_t addend)
{
#if MPG_ON_DEVICE
    uint32_t r;
    if (kHi)
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "s"(field_dword), "v"(lane_factor), "v"(addend));
    else
        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "s"(field_dword), "v"(lane_factor), "v"(addend));
    return r;
#else
    MPG_CHECK(lane_factor < 65536);
    return ((field_dword >> (kHi ? 16 : 0)) & 0xffffu) * (lane_factor & 0xffffu) + addend;
#endif
}
template <bool kHi> MPG_HD int32_t mad_i16_field(uint32_t field_dword, int32_t lane_factor, int32_t addend) // signed halves
{
#if MPG_ON_DEVICE
    int32_t r;
    if (kHi)
        asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "s"(field_dword), "v"(lane_factor), "v"(addend));
    else
        asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "s"(field_dword), "v"(lane_factor), "v"(addend));
    return r;
#else
    MPG_CHECK(lane_factor >= -32768 && lane_factor < 32768);
    return (int32_t)(int16_t)(field_dword >> (kHi ? 16 : 0)) * lane_factor + addend;
#endif
}
MPG_HD uint32_t add_u16_hi(uint32_t field_dword, uint32_t addend) // (upper half of a record dword) + addend
{
#if MPG_ON_DEVICE
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, 1, %2 op_sel:[1,0,0,0]" : "=v"(r) : "s"(field_dword), "v"(addend));
    return r;
#else
    return (field_dword >> 16) + addend;
#endif
}
MPG_HD uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) // a * b + c, a and b within 24 bits: one full-rate instruction
{
#if MPG_ON_DEVICE
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    MPG_CHECK(a < (1u << 24) && b < (1u << 24));
    return a * b + c;
#endif
}

// what depends on the lane only (worked out once per wave, while the chunk header is on its way)
struct RcLane {
    // as lane of a window load (piece `lane` < 54): luma pieces (lane < 34) are (row lane>>1, tile column lane&1),
    // chroma pieces (plane, row pair, block column).  A record's r1 / r1 + r2 is X = (offset of the tile / block that holds the
    // window origin) + 16 * (the origin's row / row pair inside it); with u = (X & sub_mask) + rj16 the piece's offset is
    // X + cterm + (u >> wrap_shift) * below: a tile / block is `wrap` bytes long, the one below it is `below + wrap` further
    // on, and u counts 16-byte pieces from the top of the first one.
    uint32_t piece_chroma; // all ones if the piece is chroma, else 0
    uint32_t sub_mask;     // 0xf0 (16 rows per tile) / 0x30 (4 row pairs per block)
    uint32_t rj16;         // 16 * (luma: row 0..16; chroma: row pair 0..4)
    uint32_t cterm[kRcMbs]; // rj16 + (tile column * 256 / plane * 64 + block column * 128) + kRcDmaBias - (LDS offset of window m): the
                           // instruction's offset field, which carries that LDS offset, is added to the global address too
    uint32_t wrap_shift;   // 8 / 6
    uint32_t below;        // mb_w * 256 - 256 / mb_w * 128 - 64
    uint32_t mc_luma;      // as MC lane (row lane>>2, quarter lane&3): LDS offset of its taps inside a window = (lane>>2)*32 + (lane&3)*4
    // chroma taps, lanes 0..31 (plane (lane>>4)&1, row (lane>>1)&7, half lane&1; lanes 32..63 mirror them: no divergence, their
    // results are not stored): the affine address map of rc_mc_chroma
    uint32_t mc_plane;     // plane * 160 — tiled chroma pieces
    uint32_t mc_c0, mc_dr, mc_dc;
    int32_t mc_cs, mc_cs2, mc_ck, mc_ck2;
    uint32_t ones;         // 0x01010101 in a vector register: the rounding byte of v_lerp_u8 (a literal would be a scalar move per use)
    uint32_t out_luma;     // where the lane's 4 luma bytes go inside O_m: lane * 4
    uint32_t out_chroma;   // 4 chroma bytes: 256 + (lane & 31) * 4
};

MPG_HD RcLane rc_lane(const VideoArgs &a, int lane)
{
    const uint32_t l = (uint32_t)lane;
    RcLane k;
    {   // (selects by an all-ones / all-zeros MASK — one v_bfi_b32 each — not `chroma ? a : b` between expressions: the compiler makes
        // divergent branches of those, a dozen scalar instructions of mask bookkeeping in every wave's prologue; and not products
        // with a 0 / 1 flag either: those become quarter-rate v_mul_lo_u32)
        const uint32_t chm = opaque(l >= 34 ? ~0u : 0u);    // a chroma piece
        const uint32_t crm = opaque(l >= 44 ? ~0u : 0u);    // ... of Cr
        auto pick = [](uint32_t mask, uint32_t if_set, uint32_t if_clear) { return (mask & if_set) | (~mask & if_clear); };
        const uint32_t cj = l - 34 - (crm & 10u);            // (chroma lanes only)
        const uint32_t rj = pick(chm, cj >> 1, l >> 1);
        k.piece_chroma = chm;
        k.sub_mask = pick(chm, 0x30u, 0xf0u);
        k.rj16 = rj * 16;
        const uint32_t cterm = rj * 16 + pick(chm, (crm & kChromaCrAt) + (cj & 1) * kChromaBlockStep, (l & 1) * 256);
#pragma unroll
        for (uint32_t m = 0; m < (uint32_t)kRcMbs; m++)
            k.cterm[m] = cterm + kRcDmaBias - (kRcWinAt + m * kRcWinBytes);
        k.wrap_shift = pick(chm, 6u, 8u);
        k.below = pick(chm, a.mb_w * kChromaBlockStep - 64, a.mb_w * 256 - 256);
    }
    k.mc_luma = (l >> 2) * 32 + (l & 3) * 4;
    k.mc_plane = ((l >> 4) & 1) * 160;
    {
        const int32_t r = (int32_t)((l >> 1) & 7), b = r & 1, h = (int32_t)(l & 1);
        k.mc_c0 = (uint32_t)((int32_t)(kRcWinLuma + k.mc_plane) + 16 * r - 8 * b + 4 * h);
        k.mc_dr = (uint32_t)(8 + 16 * b);
        k.mc_dc = (uint32_t)(4 + 8 * h);
        k.mc_ck = 8 + 16 * b;
        k.mc_ck2 = 16 - 32 * b;
        k.mc_cs = 4 + 8 * h;
        k.mc_cs2 = 8 - 16 * h;
    }
    k.ones = opaque(0x01010101u);
    k.out_luma = l * 4;
    k.out_chroma = 256 + (l & 31) * 4;
    return k;
}

// ---- step 1: the vector loads.  ONE scalar base for the chunk's words: its block words, then (n_slots dwords on) its entries
// (MPG_PROBE_ENTRY16, a TIMING-ONLY tool build — frames are wrong: lanes 2i and 2i + 1 fetch the same dword, so a wave's entry
// loads touch half the bytes, as 16-bit entries would, at one extra shift and none of the work a real 16-bit form needs to find an
// entry's block: the UPPER BOUND of what halving the entries can buy.  profiles/round6_b_*.)
#ifdef MPG_PROBE_ENTRY16
#define MPG_ENT_INDEX(at, lane) (((at) + (uint32_t)(lane)) >> 1)
#else
#define MPG_ENT_INDEX(at, lane) ((at) + (uint32_t)(lane))
#endif
MPG_HD uint32_t rc_ent_lane_offset(const RcChunk &c, uint32_t at, int lane) { return (rc_n_blocks(c) + MPG_ENT_INDEX(at, lane)) * 4; } // bytes
MPG_HD const uint32_t *rc_ent_src(const VideoArgs &a, const RcChunk &c, uint32_t at, int lane)
{
    return rc_word_base(a, c) + rc_n_blocks(c) + MPG_ENT_INDEX(at, lane); // (beyond the pass's entries: ignored; the array is padded)
}
// The five direct-to-LDS loads of a wave, issued by lanes 0..53 in this order: the table (12 pieces), windows 0..3 (54
// pieces each).  All 54 lanes take part in every one of them (one asm statement, one EXEC): the table's surplus lanes
// fetch 16 bytes that a LATER load of the same wave overwrites — loads complete in order
// (tools/microbench/lds_dma_probe3.hip) and the windows cover [kRcWinAt, kRcTileAt) completely.  Surplus lanes read
// valid memory: they run on into the next streams' tables (the table array is padded by 1 KB).
MPG_HD uint32_t rc_table_lane_offset(const RcChunk &c, int lane) { return rc_stream(c) * kRcQtabStride + (uint32_t)lane * 16; }
// the chunk's block words stay in HBM: lane (g, j) loads the word of block g of a pass when the pass starts (the first
// pass's together with the entries, before anything is waited for)
MPG_HD uint32_t rc_blk_lane_offset(uint32_t pass, int lane) { return (pass * 8 + ((uint32_t)lane >> 3)) * 4; } // bytes
// window m: scalar base = the stream's frames (rc_frame_base), lane offset = the piece's tile row.  Intra and dead macroblocks
// fetch the head of the stream's first slot, kRSlow ones the head of their reference slot: valid memory, never used (a kRSlow
// window is gathered afterwards).
MPG_HD uint32_t rc_win_offset(const RcChunk &c, int m, const RcLane &k)
{
    const uint32_t x = c.r[m][1] + (k.piece_chroma & c.r[m][2]);
    const uint32_t u = (x & k.sub_mask) + k.rj16;
    return x + mad_u24(u >> k.wrap_shift, k.below, k.cterm[m]); // (u >> wrap_shift is 0 or 1: the tile below)
}

// a kRSlow window (it leaves its plane): the reference's LINEAR reads, gathered dword by dword.  Lane < 52 = piece of the linear
// window layout: luma 17 rows x 32 bytes from the dword below the origin, then per chroma plane 9 rows x 16 bytes.  Returns the
// piece's 16 bytes.  The record names the origins as (column, row) — rc_make_record — so a dword's place is found by compares and
// shifts: a column beyond the row's end is the next row's head (the linear wrap), a luma row beyond the plane is two rows of Cb | Cr
// | pad (half as wide), a chroma row beyond Cb is Cr's, beyond Cr the pad's (which lies linearly).
MPG_HD uint32_t rc_gather_chroma_at(uint32_t mb_w, uint32_t luma_bytes, uint32_t chroma_h, uint32_t x, uint32_t t)
{
    const uint32_t in_cr = t >= chroma_h ? 1u : 0u, y = t - (in_cr ? chroma_h : 0u);
    const uint32_t tiled = luma_bytes + in_cr * kChromaCrAt + mad_u24(y >> 3, mb_w * kChromaBlockStep, (x >> 3) * kChromaBlockStep + (y & 7) * 8 + (x & 7));
    const uint32_t linear = luma_bytes + mad_u24(t, mb_w * 8, x); // (the pad, and the slack behind it: as the reference's layout has them)
    return t >= 2 * chroma_h ? linear : tiled;
}
// In LDS a gathered window is NOT laid out like the linear window (pieces of 16 bytes): dword i of piece `lane` sits at
// i * kRcGatherStride + 4 * lane — where one-dword direct-to-LDS loads put it (lane_common.h: dma4x4_to_lds).  832 of the
// window's 864 bytes.
constexpr int kRcGatherLanes = 52, kRcGatherStride = kRcGatherLanes * 4;
static_assert(4 * kRcGatherStride <= kRcWinBytes, "a gathered window fits the window's place in LDS");
// byte offsets, from the reference SLOT's start, of the four dwords of `lane`'s piece
MPG_HD void rc_gather_offsets(const VideoArgs &a, const RcChunk &c, int m, int lane, uint32_t (&at)[4])
{
    // (rare: the plane geometry from the two values the kernel keeps and the record's mb_h)
    const uint32_t mb_w = a.mb_w, luma_bytes = a.luma_bytes;
    const uint32_t luma_w = mb_w * 16, chroma_w = mb_w * 8, luma_h = c.r[m][5] * 16, chroma_h = c.r[m][5] * 8;
    const uint32_t l = (uint32_t)lane, ci = l - 34, plane = ci >= 9 ? 1u : 0u;
    const bool chroma = l >= 34;
    // (no select between two record dwords: the compiler makes an indexed load of it and moves the whole chunk to scratch)
    const uint32_t r3 = c.r[m][3], r4 = c.r[m][4];
    const uint32_t o = r3 + ((0u - (uint32_t)chroma) & (r4 - r3));
    // the piece's first dword: the origin's dword, + 16 bytes for the second luma piece of a row; its row: the origin's + the
    // lane's (a Cr piece: chroma_h rows further down the Cb | Cr | pad array)
    const uint32_t col = ((o & 0xffffu) & ~3u) + (chroma ? 0u : (l & 1) * kRcPiece);
    const uint32_t row = (o >> 16) + (chroma ? plane * chroma_h + (ci - plane * 9) : (l >> 1));
    const uint32_t w = chroma ? chroma_w : luma_w;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t x = col + 4 * (uint32_t)i, y = row;
        if (x >= w) // the linear wrap (twice: a 16-pixel-wide picture's luma piece can begin in the next row but one)
            x -= w, y++;
        if (x >= w)
            x -= w, y++;
        if (chroma) {
            at[i] = rc_gather_chroma_at(mb_w, luma_bytes, chroma_h, x, y);
        } else if (y < luma_h) {
            at[i] = mad_u24(y >> 4, mb_w * 256, (x >> 4) * 256 + (y & 15) * 16 + (x & 15));
        } else { // below the luma plane the linear reads run on into Cb | Cr | pad: two of its rows per luma-width row
            const uint32_t second = x >= chroma_w ? 1u : 0u;
            at[i] = rc_gather_chroma_at(mb_w, luma_bytes, chroma_h, x - (second ? chroma_w : 0u), (y - luma_h) * 2 + second);
        }
    }
}
// window kM of the chunk, gathered: four one-dword direct-to-LDS loads by lanes 0..51, issued BEHIND the wave's regular window
// loads (loads complete in order: what the regular load of this window fetched — the head of the reference slot — is
// written over) and waited for with them.  Until round 6 the gather ran when the motion compensation got to the macroblock:
// four dependent loads into registers, a wait, a round trip through LDS — per such macroblock, one after the other.
template <int kM> MPG_HD void rc_gather_to_lds(const VideoArgs &a, const RcChunk &c, const uint8_t *fbase, uint8_t *lds, int lane)
{
    constexpr int kAt = kRcWinAt + kM * kRcWinBytes;
    uint32_t at[4], off[4];
    rc_gather_offsets(a, c, kM, lane, at);
#pragma unroll
    for (int i = 0; i < 4; i++) // (fbase sits kRcDmaBias below the stream's frames; the instruction adds its offset field to the global address too)
        off[i] = kRcDmaBias + c.r[kM][1] + at[i] - (uint32_t)(kAt + i * kRcGatherStride);
    dma4x4_to_lds<kAt, kRcGatherStride>(fbase, off, lds, lane);
}
MPG_HD bool rc_any_slow(const RcChunk &c) { return ((c.r[0][0] | c.r[1][0] | c.r[2][0] | c.r[3][0]) & kRSlow) != 0; }

// ---- step 2: the residual pass
struct __attribute__((packed, aligned(4))) i32x4_a4 { int32_t v[4]; }; // 16 bytes at dword alignment (the words array)
// (byte kByte of `bytes`) * (half-word kHalf of `words`, sign-extended): unpacking is the multiplier's operand select
template <int kByte, int kHalf> MPG_HD int32_t mul_u8_s16(uint32_t bytes, uint32_t words)
{
    static_assert((kByte & 1) == kHalf, "rows 2k, 2k+1 of a column: bytes 2k, 2k+1 and the two halves of word k");
#if MPG_ON_DEVICE
    int32_t r;
    if (kByte == 0 && kHalf == 0)
        asm("v_mul_i32_i24_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0" : "=v"(r) : "v"(bytes), "v"(words));
    else if (kByte == 1 && kHalf == 1)
        asm("v_mul_i32_i24_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_1" : "=v"(r) : "v"(bytes), "v"(words));
    else if (kByte == 2 && kHalf == 0)
        asm("v_mul_i32_i24_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_0" : "=v"(r) : "v"(bytes), "v"(words));
    else
        asm("v_mul_i32_i24_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:WORD_1" : "=v"(r) : "v"(bytes), "v"(words));
    return r;
#else
    return (int32_t)((bytes >> (8 * kByte)) & 0xff) * (int32_t)(int16_t)(words >> (16 * kHalf));
#endif
}
// x (a small non-negative dword) * half kHalf of `words` (sign-extended), one instruction
template <int kHalf> MPG_HD int32_t mul_s16(uint32_t x, uint32_t words)
{
#if MPG_ON_DEVICE
    int32_t r;
    if (kHalf == 0)
        asm("v_mul_i32_i24_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(x), "v"(words));
    else
        asm("v_mul_i32_i24_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(x), "v"(words));
    return r;
#else
    MPG_CHECK(x < (1u << 23));
    return (int32_t)x * (int32_t)(int16_t)(words >> (16 * kHalf));
#endif
}
MPG_HD void rc_zero_tile16(int16_t *T, int lane)
{
    const i32x4 z = {{0, 0, 0, 0}};
    *reinterpret_cast<i32x4 *>(T + lane * 8) = z;
}
// the int16 tile: the dequantised level (video.go:719-741) to T[slot & 7][position], |.| <= 2048
MPG_HD void rc_scatter16(int16_t *T, const uint8_t *lds, uint32_t e)
{
    const uint8_t *Q = lds + kRcQtabAt;
    const int32_t qm = Q[(e & 0xfeu) >> 1];
    const int32_t level = (int32_t)e >> 16;
    const int32_t qs = (int32_t)((e >> 11) & 31);
    T[(e & 0x7fcu) >> 2] = (int16_t)dequant_level(level, !(e & kENonIntra), mul24_as_written(qs, qm));
}
// an intra block's DC (video.go:656-672: `<<= 3 + 5`, no matrix), from its block word: lane (g, 0), row 0 of column 0
MPG_HD void rc_dc_from_word(uint32_t bw, int lane, int32_t (&v)[8])
{
    if ((bw & kBDcWord) && (lane & 7) == 0)
        v[0] = ((int32_t)(bw << 4) >> 16) * 256;
}
// lane (g, j): column j of block g from the int16 tile, premultiplied (video.go:744)
MPG_HD void rc_cols_load16(const int16_t *T, const uint8_t *lds, int lane, int32_t (&v)[8])
{
    const u32x4 t = *reinterpret_cast<const u32x4 *>(T + lane * 8); // T[g][j * 8 + r]: rows 0..7 of column j
    const uint32_t *pmp = reinterpret_cast<const uint32_t *>(lds + kRcQtabAt + 128 + ((uint32_t)lane & 7) * 8);
    const uint32_t p0 = pmp[0], p1 = pmp[1];
    v[0] = mul_u8_s16<0, 0>(p0, t.v[0]);
    v[1] = mul_u8_s16<1, 1>(p0, t.v[0]);
    v[2] = mul_u8_s16<2, 0>(p0, t.v[1]);
    v[3] = mul_u8_s16<3, 1>(p0, t.v[1]);
    v[4] = mul_u8_s16<0, 0>(p1, t.v[2]);
    v[5] = mul_u8_s16<1, 1>(p1, t.v[2]);
    v[6] = mul_u8_s16<2, 0>(p1, t.v[3]);
    v[7] = mul_u8_s16<3, 1>(p1, t.v[3]);
}
// an int32 snapshot block: lane (g, j) takes column j (positions j * 8 .. j * 8 + 7) straight from its 64 dwords
MPG_HD void rc_raw_cols(const VideoArgs &a, const RcChunk &c, uint32_t bw, int lane, int32_t (&v)[8])
{
    const uint32_t at = (rc_n_blocks(c) + ((bw >> 12) & 0xfffu) + ((uint32_t)lane & 7) * 8) * 4; // (bytes, 32 bits: rc_dense_read)
    const i32x4_a4 *p = reinterpret_cast<const i32x4_a4 *>(reinterpret_cast<const uint8_t *>(rc_word_base(a, c)) + at);
    const i32x4_a4 lo = p[0], hi = p[1];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        v[r] = lo.v[r];
        v[r + 4] = hi.v[r];
    }
}
// 8 x 8 transposition across the 8 lanes of a block: lane j holds column j (v[r] = row r) and leaves with row j (v[c] =
// column c).  Three exchange steps (lane bit k against register bit k); the partner's value comes by DPP.  (Device only:
// the emulator, which runs lane after lane, transposes at the wave level.)
#if MPG_ON_DEVICE
// One exchange step inside quads, written out: 8 selects that take their other operand through DPP (v_cndmask_b32_dpp: D =
// VCC ? src1 : dpp(src0)) instead of the 8 DPP moves + 8 selects the compiler makes of the same thing (28 instead of 44 instructions per transposition:
// profiles/r5_ab_*: +2 % typical).  a[i] / b[i]: the four
// register pairs of this step (bit clear / set), `set_lanes`: the lanes whose bit is set.
#define MPG_QUAD_STEP(PERM)                                                                                                          \
    asm volatile("s_nop 1\n\t"                                                                                                       \
                 "s_mov_b64 vcc, %[set]\n\t"                                                                                         \
                 "v_cndmask_b32_dpp %[nb0], %[a0], %[b0], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[nb1], %[a1], %[b1], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[nb2], %[a2], %[b2], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[nb3], %[a3], %[b3], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "s_not_b64 vcc, vcc\n\t"                                                                                            \
                 "v_cndmask_b32_dpp %[na0], %[b0], %[a0], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[na1], %[b1], %[a1], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[na2], %[b2], %[a2], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[na3], %[b3], %[a3], vcc " PERM " row_mask:0xf bank_mask:0xf"                                     \
                 : [na0] "=&v"(na[0]), [na1] "=&v"(na[1]), [na2] "=&v"(na[2]), [na3] "=&v"(na[3]), [nb0] "=&v"(nb[0]),                \
                   [nb1] "=&v"(nb[1]), [nb2] "=&v"(nb[2]), [nb3] "=&v"(nb[3])                                                         \
                 : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]),    \
                   [b3] "v"(b[3]), [set] "s"(set_lanes)                                                                               \
                 : "vcc", "scc")
#endif
MPG_HD void rc_transpose8(int32_t (&v)[8], int lane)
{
#if MPG_ON_DEVICE
    (void)lane;
    {   // lane bit 0 <-> register bit 0: pairs (0,1) (2,3) (4,5) (6,7); partner = lane ^ 1
        const int32_t a[4] = {v[0], v[2], v[4], v[6]}, b[4] = {v[1], v[3], v[5], v[7]};
        int32_t na[4], nb[4];
        const uint64_t set_lanes = 0xAAAAAAAAAAAAAAAAull;
        MPG_QUAD_STEP("quad_perm:[1,0,3,2]");
        v[0] = na[0], v[2] = na[1], v[4] = na[2], v[6] = na[3];
        v[1] = nb[0], v[3] = nb[1], v[5] = nb[2], v[7] = nb[3];
    }
    {   // bit 1: pairs (0,2) (1,3) (4,6) (5,7); partner = lane ^ 2
        const int32_t a[4] = {v[0], v[1], v[4], v[5]}, b[4] = {v[2], v[3], v[6], v[7]};
        int32_t na[4], nb[4];
        const uint64_t set_lanes = 0xCCCCCCCCCCCCCCCCull;
        MPG_QUAD_STEP("quad_perm:[2,3,0,1]");
        v[0] = na[0], v[1] = na[1], v[4] = na[2], v[5] = na[3];
        v[2] = nb[0], v[3] = nb[1], v[6] = nb[2], v[7] = nb[3];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { // bit 2: lanes 0..3 of a block take from lane + 4 (banks 0, 2), lanes 4..7 from lane - 4
        const int32_t a = v[r], b = v[r + 4];
        v[r + 4] = __builtin_amdgcn_update_dpp(b, a, 0x104, 0xf, 0x5, false);
        v[r] = __builtin_amdgcn_update_dpp(a, b, 0x114, 0xf, 0xa, false);
    }
#else
    (void)v;
    (void)lane;
#endif
}

// a dense block (more than 32 non-zero levels): lane (g, j) takes column j straight from the unit — one 16-byte load, 8
// levels dequantised in place of the tile read.  The column's matrix entries (of the lane's class) and premultipliers
// are brought into two dwords each, so that every product takes its byte operand through the multiplier's own byte
// select (SDWA) instead of a shift and a mask.
template <int kByte> MPG_HD int32_t mul_u8(uint32_t bytes, int32_t x) // (byte kByte of `bytes`) * x, both within 24 bits
{
#if MPG_ON_DEVICE
    int32_t r;
    if (kByte == 0)
        asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(x), "v"(bytes));
    else if (kByte == 1)
        asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(x), "v"(bytes));
    else if (kByte == 2)
        asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(x), "v"(bytes));
    else
        asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(x), "v"(bytes));
    return r;
#else
    MPG_CHECK(x >= -(1 << 23) && x < (1 << 23));
    return (int32_t)((bytes >> (8 * kByte)) & 0xff) * x;
#endif
}
// bytes 0, 2 (or 1, 3 for the odd class) of lo and of hi -> one dword
MPG_HD uint32_t pick_class_bytes(uint32_t lo, uint32_t hi, bool odd)
{
#if MPG_ON_DEVICE
    return __builtin_amdgcn_perm(hi, lo, odd ? 0x07050301u : 0x06040200u);
#else
    const uint32_t s = odd ? 8 : 0;
    return ((lo >> s) & 0xff) | (((lo >> (16 + s)) & 0xff) << 8) | (((hi >> s) & 0xff) << 16) | (((hi >> (16 + s)) & 0xff) << 24);
#endif
}
// ---- a dense unit's levels, two at a time (rows 2k, 2k + 1 of a column = the two halves of one dword).  video.go:719-744
// on PACKED halves as far as it goes: sign(level) in {-1, 0, +1} (0 for intra blocks: no "+ sign") and u = 2 level +
// sign(level) by three packed 16-bit instructions per pair; then per level (u, sign-extended by the multiplier's operand
// select) * matrix byte * quantiser_scale >> 4, "if even, one toward zero" as l + ((0 - l) >> 31), and `| (level != 0)` in the
// place of the reference's `| 1`: a zero level (not coded: the reference leaves the coefficient alone) then comes out
// as 0 through the whole chain — u = 0, product 0, 0 | 0 — with no select at the end.  u must fit int16: the packer sends
// blocks with a level beyond +-16383 as sparse entries (kRcDenseLevelMax; the entry path works in 32 bits).
struct RcPair { uint32_t u, nz; }; // halves: 2 level + sign(level) / level != 0
MPG_HD RcPair rc_dense_pair(uint32_t w, uint32_t non_intra_mask)
{
    RcPair p;
#if MPG_ON_DEVICE
    // (written out: the compiler makes compares and selects of the packed minimum, and a shift + add of the multiply-add)
    uint32_t sg;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(p.nz) : "v"(w));
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(sg) : "v"(w));
    sg = (sg | p.nz) & non_intra_mask;
    asm("v_pk_mad_i16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(p.u) : "v"(w), "v"(sg));
#else
    p.u = p.nz = 0;
    for (int h = 0; h < 2; h++) {
        const int32_t level = (int16_t)(w >> (16 * h));
        const int32_t sg = non_intra_mask ? (level > 0) - (level < 0) : 0;
        p.nz |= (uint32_t)(level != 0) << (16 * h);
        p.u |= (uint32_t)(uint16_t)(2 * level + sg) << (16 * h); // (wraps beyond +-16383: the packer keeps those out)
    }
#endif
    return p;
}
template <int kHalf> MPG_HD int32_t or_half(int32_t x, uint32_t halves) // x | (half kHalf of `halves`, zero-extended)
{
#if MPG_ON_DEVICE
    int32_t r;
    if (kHalf == 0)
        asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(x), "v"(halves));
    else
        asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(x), "v"(halves));
    return r;
#else
    return x | (int32_t)((halves >> (16 * kHalf)) & 0xffff);
#endif
}
// the chain behind the product: "if even, one toward zero", `| 1`, clamp, premultiply
template <int R> MPG_HD int32_t rc_dense_finish(int32_t l, const RcPair &p, const uint32_t (&pm)[2])
{
    l += (int32_t)opaque((uint32_t)(0 - l)) >> 31; // l > 0: one down — then odd, or one below an odd number.  (Three plain
                                             // half-cost instructions; the compiler's own form is a compare into a scalar
                                             // pair and a subtract-with-borrow, two full-cost ones and a hazard nop.)
    l = or_half<R & 1>(l, p.nz);             // "| 1" of a coded level; 0 stays 0
    l = clampi(l, -2048, 2047);
    return mul_u8<R & 3>(pm[R >> 2], l);
}
template <int R> MPG_HD int32_t rc_dense_level(const RcPair &p, int32_t qs, const uint32_t (&qm)[2], const uint32_t (&pm)[2])
{
    const int32_t l = mul24_as_written(mul_u8_s16<R & 3, R & 1>(qm[R >> 2], p.u), qs) >> 4; // |u * matrix byte| < 2^23
    return rc_dense_finish<R>(l, p, pm);
}
// the same where the matrix entry is 16 — the default non-intra matrix (video.go:1066-1075), i.e. nearly every stream's:
// (u * 16 * quantiser_scale) >> 4 = u * quantiser_scale exactly, one multiply in the place of two and a shift
template <int R> MPG_HD int32_t rc_dense_level_flat(const RcPair &p, int32_t qs, const uint32_t (&pm)[2])
{
    return rc_dense_finish<R>(mul_s16<R & 1>((uint32_t)qs, p.u), p, pm);
}
// ---- the same on PACKED halves from end to end (round 4).  With the matrix entry 16 the reference's chain collapses:
// u = 2 level + sign is odd, so l = u * quantiser_scale is even exactly when quantiser_scale is, and "one toward zero if
// even" takes sign(level) off it then: l = A level + e sign(level) with A = 2 quantiser_scale, e = (quantiser_scale - 1) | 1
// — odd by construction (the `| 1` has nothing left to do), 0 for a level of 0, and within int16 for levels up to
// kRcDenseLevelMax.  Per PAIR of levels: sign (2 packed instructions), A level (1), + e sign (1), the clamp to
// [-2048, 2047] (2), then the two premultiplies, whose operand select unpacks the halves: 8 instructions where the 32-bit
// form took 18 (33 -> 17 issue clocks per level, profiles/r02j_valu_issue_rates.txt).
MPG_HD uint32_t pk_sign_i16(uint32_t w) // per half: -1, 0, +1
{
#if MPG_ON_DEVICE
    uint32_t t, r;
    asm("v_pk_min_i16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(t) : "v"(w));
    asm("v_pk_max_i16 %0, %1, -1 op_sel_hi:[1,0]" : "=v"(r) : "v"(t));
    return r;
#else
    uint32_t r = 0;
    for (int h = 0; h < 2; h++) {
        const int32_t x = (int16_t)(w >> (16 * h));
        r |= (uint32_t)(uint16_t)((x > 0) - (x < 0)) << (16 * h);
    }
    return r;
#endif
}
MPG_HD uint32_t pk_mul_lo_i16(uint32_t a, uint32_t b) // per half: a * b (callers keep it within int16)
{
#if MPG_ON_DEVICE
    uint32_t r;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    uint32_t r = 0;
    for (int h = 0; h < 2; h++) {
        const int32_t x = (int32_t)(int16_t)(a >> (16 * h)) * (int32_t)(int16_t)(b >> (16 * h));
        MPG_CHECK(x >= -32768 && x <= 32767);
        r |= (uint32_t)(uint16_t)x << (16 * h);
    }
    return r;
#endif
}
MPG_HD uint32_t pk_mad_i16(uint32_t a, uint32_t b, uint32_t c) // per half: a * b + c (within int16)
{
#if MPG_ON_DEVICE
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    uint32_t r = 0;
    for (int h = 0; h < 2; h++) {
        const int32_t x = (int32_t)(int16_t)(a >> (16 * h)) * (int32_t)(int16_t)(b >> (16 * h)) + (int32_t)(int16_t)(c >> (16 * h));
        MPG_CHECK(x >= -32768 && x <= 32767);
        r |= (uint32_t)(uint16_t)x << (16 * h);
    }
    return r;
#endif
}
MPG_HD uint32_t pk_clamp12_i16(uint32_t x, uint32_t lo2, uint32_t hi2) // per half: clamp(x, -2048, 2047); lo2 / hi2: the bounds in both halves
{
#if MPG_ON_DEVICE
    uint32_t t, r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(t) : "v"(x), "s"(lo2));
    asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(t), "s"(hi2));
    return r;
#else
    (void)lo2;
    (void)hi2;
    uint32_t r = 0;
    for (int h = 0; h < 2; h++)
        r |= (uint32_t)(uint16_t)clampi((int16_t)(x >> (16 * h)), -2048, 2047) << (16 * h);
    return r;
#endif
}
// a pair of levels (rows 2k, 2k + 1 of the lane's column) -> their dequantised, clamped values in the two halves
#ifndef MPG_DENSE_SAT16
#define MPG_DENSE_SAT16 1
#endif
MPG_HD uint32_t pk_mad_sat_i16(uint32_t a, uint32_t b, uint32_t c) // per half: saturate_int16(a * b + c), the sum in full precision
{
#if MPG_ON_DEVICE
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    uint32_t r = 0;
    for (int h = 0; h < 2; h++) {
        const int32_t x = (int32_t)(int16_t)(a >> (16 * h)) * (int32_t)(int16_t)(b >> (16 * h)) + (int32_t)(int16_t)(c >> (16 * h));
        r |= (uint32_t)(uint16_t)clampi(x, -32768, 32767) << (16 * h);
    }
    return r;
#endif
}
MPG_HD uint32_t pk_ashr4_i16(uint32_t x)
{
#if MPG_ON_DEVICE
    uint32_t r;
    asm("v_pk_ashrrev_i16 %0, 4, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(x));
    return r;
#else
    uint32_t r = 0;
    for (int h = 0; h < 2; h++)
        r |= (uint32_t)(uint16_t)((int16_t)(x >> (16 * h)) >> 4) << (16 * h);
    return r;
#endif
}
// a2 / e2: 16 A / 16 e in both halves.  The clamp to [-2048, 2047] is the multiply-add's own saturation at sixteen times the
// scale — 16 l saturates at +-32767 / -32768, whose sixteenth (an arithmetic shift) is 2047 / -2048 — one instruction in the
// place of a packed maximum and a packed minimum (MPG_DENSE_SAT16 0: those).
MPG_HD uint32_t rc_dense_pair_flat(uint32_t w, uint32_t a2, uint32_t e2)
{
    const uint32_t sg = pk_sign_i16(w);
#if MPG_DENSE_SAT16
    return pk_ashr4_i16(pk_mad_sat_i16(w, a2, pk_mul_lo_i16(sg, e2)));
#else
    const uint32_t l = pk_mad_i16(sg, e2, pk_mul_lo_i16(w, a2));
    return pk_clamp12_i16(l, 0xf800f800u, 0x07ff07ffu);
#endif
}

// Is column j = lane & 7 of the stream's non-intra matrix all 16?  (The wave's AND over its lanes answers for the matrix.)
MPG_HD bool rc_non_intra_column_flat(const uint8_t *lds, int lane)
{
    const u32x4 q = *reinterpret_cast<const u32x4 *>(lds + kRcQtabAt + ((uint32_t)lane & 7) * 16); // bytes 2r + 1: non-intra
    const uint32_t differs = (q.v[0] ^ 0x10001000u) | (q.v[1] ^ 0x10001000u) | (q.v[2] ^ 0x10001000u) | (q.v[3] ^ 0x10001000u);
    return (differs & 0xff00ff00u) == 0;
}
// column j of a dense unit: 8 int16 levels (read apart from their use: the next pass's are fetched while this pass runs)
MPG_HD i32x4_a4 rc_dense_read(const VideoArgs &a, const RcChunk &c, uint32_t bw, int lane)
{
    // (the BYTE offset as a 32-bit value on top of the wave's scalar base: written as a dword index the compiler widens it per lane
    // — two 64-bit shift-adds in front of every unit load)
    const uint32_t at = (rc_n_blocks(c) + ((bw >> 12) & 0xfffu) + ((uint32_t)lane & 7) * 4) * 4;
    return *reinterpret_cast<const i32x4_a4 *>(reinterpret_cast<const uint8_t *>(rc_word_base(a, c)) + at);
}
// kFlat: every dense block of the pass is non-intra and the stream's non-intra matrix is 16 everywhere (the caller's
// wave-uniform test): no matrix bytes, no intra DC.
template <bool kFlat = false>
MPG_HD void rc_dense_cols(const i32x4_a4 &lv, const uint8_t *lds, uint32_t bw, int lane, int32_t (&v)[8])
{
    const uint32_t j = (uint32_t)lane & 7;
    const int32_t qs = (int32_t)((bw >> 26) & 31);
    const bool intra = !(bw >> 31);
    const uint32_t *pmp = reinterpret_cast<const uint32_t *>(lds + kRcQtabAt + 128 + j * 8); // the column's 8 premultipliers
    const uint32_t pm[2] = {pmp[0], pmp[1]};
#if !MPG_ON_DEVICE
    for (int r = (intra && j == 0) ? 1 : 0; r < 8; r++) {
        const int32_t level = (int16_t)((uint32_t)lv.v[r >> 1] >> (16 * (r & 1)));
        MPG_CHECK(level >= -kRcDenseLevelMax && level <= kRcDenseLevelMax);
        (void)level;
    }
#endif
    if (kFlat) {
        MPG_CHECK(!intra && rc_non_intra_column_flat(lds, lane));
        const uint32_t a1 = (uint32_t)qs << (MPG_DENSE_SAT16 ? 5 : 1), e1 = (((uint32_t)qs - 1) | 1) << (MPG_DENSE_SAT16 ? 4 : 0); // A, e of the block (x 16); in both halves:
        const uint32_t a2 = a1 | (a1 << 16), e2 = e1 | (e1 << 16);
#if MPG_DENSE_SAT16
        // (the four pairs stage by stage, not pair by pair: a packed instruction that consumes the result of the one in front of it
        // costs a wait state — the compiler's own order had eight s_nop per column)
        const uint32_t w0 = (uint32_t)lv.v[0], w1 = (uint32_t)lv.v[1], w2 = (uint32_t)lv.v[2], w3 = (uint32_t)lv.v[3];
        const uint32_t g0 = pk_sign_i16(w0), g1 = pk_sign_i16(w1), g2 = pk_sign_i16(w2), g3 = pk_sign_i16(w3);
        const uint32_t m0 = pk_mul_lo_i16(g0, e2), m1 = pk_mul_lo_i16(g1, e2), m2 = pk_mul_lo_i16(g2, e2), m3 = pk_mul_lo_i16(g3, e2);
        const uint32_t s0 = pk_mad_sat_i16(w0, a2, m0), s1 = pk_mad_sat_i16(w1, a2, m1), s2 = pk_mad_sat_i16(w2, a2, m2), s3 = pk_mad_sat_i16(w3, a2, m3);
        const uint32_t l0 = pk_ashr4_i16(s0), l1 = pk_ashr4_i16(s1), l2 = pk_ashr4_i16(s2), l3 = pk_ashr4_i16(s3);
#else
        const uint32_t l0 = rc_dense_pair_flat((uint32_t)lv.v[0], a2, e2), l1 = rc_dense_pair_flat((uint32_t)lv.v[1], a2, e2);
        const uint32_t l2 = rc_dense_pair_flat((uint32_t)lv.v[2], a2, e2), l3 = rc_dense_pair_flat((uint32_t)lv.v[3], a2, e2);
#endif
        v[0] = mul_u8_s16<0, 0>(pm[0], l0); // the premultiply (video.go:744): byte r of the column's eight x half r & 1
        v[1] = mul_u8_s16<1, 1>(pm[0], l0);
        v[2] = mul_u8_s16<2, 0>(pm[0], l1);
        v[3] = mul_u8_s16<3, 1>(pm[0], l1);
        v[4] = mul_u8_s16<0, 0>(pm[1], l2);
        v[5] = mul_u8_s16<1, 1>(pm[1], l2);
        v[6] = mul_u8_s16<2, 0>(pm[1], l3);
        v[7] = mul_u8_s16<3, 1>(pm[1], l3);
        return;
    }
    // the column's 8 matrix entries of both classes (16 bytes: position j * 8 + r -> bytes 2r, 2r + 1)
    const u32x4 q = *reinterpret_cast<const u32x4 *>(lds + kRcQtabAt + j * 16);
    const uint32_t qm[2] = {pick_class_bytes(q.v[0], q.v[1], !intra), pick_class_bytes(q.v[2], q.v[3], !intra)};
    const uint32_t non_intra_mask = (uint32_t)((int32_t)bw >> 31);
    const RcPair p0 = rc_dense_pair((uint32_t)lv.v[0], non_intra_mask), p1 = rc_dense_pair((uint32_t)lv.v[1], non_intra_mask);
    const RcPair p2 = rc_dense_pair((uint32_t)lv.v[2], non_intra_mask), p3 = rc_dense_pair((uint32_t)lv.v[3], non_intra_mask);
    v[0] = rc_dense_level<0>(p0, qs, qm, pm);
    v[1] = rc_dense_level<1>(p0, qs, qm, pm);
    v[2] = rc_dense_level<2>(p1, qs, qm, pm);
    v[3] = rc_dense_level<3>(p1, qs, qm, pm);
    v[4] = rc_dense_level<4>(p2, qs, qm, pm);
    v[5] = rc_dense_level<5>(p2, qs, qm, pm);
    v[6] = rc_dense_level<6>(p3, qs, qm, pm);
    v[7] = rc_dense_level<7>(p3, qs, qm, pm);
    if (intra && j == 0)
        v[0] = (int32_t)(int16_t)(lv.v[0] & 0xffff) * 256; // DC: `<<= 3+5`, video.go:672
}

// The transposition through LDS (kT16 = false), one half of the wave at a time: T = [4 blocks][kRcTposeStride] int32 over the
// tile.  Lane (g, j) stores column j of block g & 3 (v[r] = row r) and loads row j back (v[c] = column c).  The caller runs
// lanes 0..31, then lanes 32..63, store before load (mpeghip.hip: rc_transpose8_lds; the emulator the same, lane after lane).
MPG_HD void rc_tpose_store(int32_t *T, int lane, const int32_t (&v)[8])
{
    int32_t *t = T + ((lane >> 3) & 3) * kRcTposeStride + (lane & 7);
#pragma unroll
    for (int r = 0; r < 8; r++)
        t[r * 8] = v[r];
}
MPG_HD void rc_tpose_load(const int32_t *T, int lane, int32_t (&v)[8])
{
    const i32x4 *t = reinterpret_cast<const i32x4 *>(T + ((lane >> 3) & 3) * kRcTposeStride + (lane & 7) * 8);
    const i32x4 t0 = t[0], t1 = t[1];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        v[r] = t0.v[r];
        v[r + 4] = t1.v[r];
    }
}

// ---- step 3: motion compensation of 4 pixels (video_noasm.go:48-80); shifts / oh / ov are wave-uniform.
// a0 a1: the two dwords that hold the pixels (they start sh / 8 < 4 bytes in) and their right neighbour; b0 b1 the same one
// row below.  sh: funnel shift in bits (its low 5 bits count), sh8: sh + 8 (its low 6 bits count) — scalars the packer worked
// out (record r3 / r4), or rc_slow_shifts for a gathered window.  ones: 0x01010101.
MPG_HD uint32_t funnel32(uint32_t hi, uint32_t lo, uint32_t sh) // ({hi, lo} >> (sh & 31)): one v_alignbit_b32, no masking of sh
{
#if MPG_ON_DEVICE
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31));
#endif
}
MPG_HD uint32_t avg_ceil_u8x4_r(uint32_t a, uint32_t b, uint32_t ones)
{
#if MPG_ON_DEVICE
    return __builtin_amdgcn_lerp(a, b, ones);
#else
    (void)ones;
    return avg_ceil_u8x4(a, b);
#endif
}
MPG_HD uint32_t avg4_u8x4_r(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t ones)
{
    const uint32_t p = avg_floor_u8x4(a, b), q = avg_floor_u8x4(c, d);
    const uint32_t e = (a ^ b) & (c ^ d);
    return avg_ceil_u8x4_r(p, q, ones) + (e & ~(p ^ q) & ones);
}
MPG_HD uint32_t rc_mc4_h(uint32_t a0, uint32_t a1, uint32_t sh, uint32_t sh8, uint32_t ones)
{
    const uint64_t a = (uint64_t)a0 | ((uint64_t)a1 << 32);
    return avg_ceil_u8x4_r(funnel32(a1, a0, sh), (uint32_t)(a >> (sh8 & 63)), ones);
}
MPG_HD uint32_t rc_mc4_v(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t sh, uint32_t ones)
{
    return avg_ceil_u8x4_r(funnel32(a1, a0, sh), funnel32(b1, b0, sh), ones);
}
MPG_HD uint32_t rc_mc4_hv(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t sh, uint32_t sh8, uint32_t ones)
{
    const uint64_t a = (uint64_t)a0 | ((uint64_t)a1 << 32), b = (uint64_t)b0 | ((uint64_t)b1 << 32);
    return avg4_u8x4_r(funnel32(a1, a0, sh), (uint32_t)(a >> (sh8 & 63)), funnel32(b1, b0, sh), (uint32_t)(b >> (sh8 & 63)), ones);
}

// Two / four dwords of the wave's LDS at (lane part) + (compile-time part) each: the compile-time part rides in the instruction's
// 16-bit offset field, and the wait sits in the same statement, so the values are there when it ends.  (Left to itself the
// compiler pairs such reads into ds_read2_b32, whose offsets reach 1 KB: behind window 0 that costs an address addition per
// pair; two ds_read_b32 are no slower — profiles/round5_b_lds_unaligned_reads.txt.)
template <int kAt0, int kAt1> MPG_HD void lds_read32x2(const uint8_t *lds, uint32_t p0, uint32_t p1, uint32_t &v0, uint32_t &v1)
{
#if MPG_ON_DEVICE
    const uint32_t base = (uint32_t)(uintptr_t)lds; // (one wave per workgroup: 0)
    asm volatile("ds_read_b32 %0, %2 offset:%4\n\tds_read_b32 %1, %3 offset:%5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1)
                 : "v"(p0 + base), "v"(p1 + base), "n"(kAt0), "n"(kAt1)
                 : "memory");
#else
    memcpy(&v0, lds + p0 + kAt0, 4);
    memcpy(&v1, lds + p1 + kAt1, 4);
#endif
}
template <int kAt0, int kAt1, int kAt2, int kAt3>
MPG_HD void lds_read32x4(const uint8_t *lds, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t &v0, uint32_t &v1, uint32_t &v2, uint32_t &v3)
{
#if MPG_ON_DEVICE
    const uint32_t base = (uint32_t)(uintptr_t)lds;
    asm volatile("ds_read_b32 %0, %4 offset:%8\n\tds_read_b32 %1, %5 offset:%9\n\tds_read_b32 %2, %6 offset:%10\n\tds_read_b32 %3, %7 offset:%11\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                 : "v"(p0 + base), "v"(p1 + base), "v"(p2 + base), "v"(p3 + base), "n"(kAt0), "n"(kAt1), "n"(kAt2), "n"(kAt3)
                 : "memory");
#else
    memcpy(&v0, lds + p0 + kAt0, 4);
    memcpy(&v1, lds + p1 + kAt1, 4);
    memcpy(&v2, lds + p2 + kAt2, 4);
    memcpy(&v3, lds + p3 + kAt3, 4);
#endif
}

// ---- the common case: a window inside its plane, tiled (record r3 / r4 / r5 as rc_make_record packs them).
// luma, lane (row lane>>2, quarter lane&3): rows are 32 contiguous bytes in window kM; the lane's taps start r3's upper half
// bytes (0 / 4 / 8 / 12) into its dword pair.
template <int kM> MPG_HD uint32_t rc_mc_luma(const uint8_t *lds, const RcLane &k, uint32_t r0, uint32_t r3)
{
    constexpr int W = kRcWinAt + kM * kRcWinBytes;
    const uint32_t at = add_u16_hi(r3, k.mc_luma);
    uint32_t a0, a1, b0, b1;
    if (r0 & kROvL) {
        lds_read32x4<W, W + 4, W + 32, W + 36>(lds, at, at, at, at, a0, a1, b0, b1); // (+ 32 bytes: the row below)
        if (r0 & kROhL)
            return rc_mc4_hv(a0, a1, b0, b1, r3, r3 >> 8, k.ones);
        return rc_mc4_v(a0, a1, b0, b1, r3, k.ones);
    }
    lds_read32x2<W, W + 4>(lds, at, at, a0, a1);
    if (r0 & kROhL)
        return rc_mc4_h(a0, a1, r3, r3 >> 8, k.ones);
    return funnel32(a1, a0, r3);
}

// chroma, lanes 0..31 (plane lane>>4, row r = (lane>>1)&7, half h = lane&1; lanes 32..63 repeat them).  A row's 16 bytes are two
// halves of 8 in the pieces of the two blocks: row w of the window (counted from the even row its first pair starts on) begins
// at rowpart(w) = (w >> 1) * 32 + (w & 1) * 8, and dword i (0..3) of a row sits at f(i) = (i & 1) * 4 + (i >> 1) * 16.  The lane
// reads dwords i, i + 1 of rows w, w + 1 with w = r + c, i = s + h, where c (0 / 1: the window starts on the odd row of its first
// pair, r4's upper half) and s (0 / 1: in the second dword, r5) are wave-uniform.  Both maps are affine in c and s for a FIXED lane:
//      rowpart(r + c)     = (16 r - 8 b)  + c (8 + 16 b)                 b = r & 1
//      rowpart(r + c + 1) = rowpart(r + c) + (8 + 16 b) + c (16 - 32 b)
//      f(s + h)           = 4 h           + s (4 + 8 h)
//      f(s + h + 1)       = f(s + h) + (4 + 8 h) + s (8 - 16 h)
// so the addresses cost multiply-adds of a record half by lane constants — no scalar instruction, no shifts and masks.
MPG_HD uint32_t rc_chroma_row_at(uint32_t plane160, uint32_t row, uint32_t dword)
{   // the closed form the affine one is checked against (tests/kernel_emu): LDS offset of dword `dword` of row `row`
    return kRcWinLuma + plane160 + (row >> 1) * 32 + (row & 1) * 8 + (dword & 1) * 4 + (dword >> 1) * 16;
}
template <int kM> MPG_HD uint32_t rc_mc_chroma(const uint8_t *lds, const RcLane &k, int lane, uint32_t r0, uint32_t r4, uint32_t r5)
{
    constexpr int W = kRcWinAt + kM * kRcWinBytes;
    (void)lane; // (the emulator's checks below)
    const uint32_t at0 = mad_u16_field<false>(r5, k.mc_dc, mad_u16_field<true>(r4, k.mc_dr, k.mc_c0));       // row w, dword i
    const uint32_t at1 = at0 + (uint32_t)mad_i16_field<false>(r5, k.mc_cs2, k.mc_cs);                          // row w, dword i + 1
    MPG_CHECK(at0 == rc_chroma_row_at(k.mc_plane, ((r4 >> 16) & 1) + (((uint32_t)lane >> 1) & 7), (r5 & 1) + ((uint32_t)lane & 1)));
    MPG_CHECK(at1 == rc_chroma_row_at(k.mc_plane, ((r4 >> 16) & 1) + (((uint32_t)lane >> 1) & 7), (r5 & 1) + ((uint32_t)lane & 1) + 1));
    uint32_t a0, a1, b0, b1;
    if (r0 & kROvC) {
        const uint32_t down = (uint32_t)mad_i16_field<true>(r4, k.mc_ck2, k.mc_ck); // to the row below
        MPG_CHECK(at0 + down == rc_chroma_row_at(k.mc_plane, ((r4 >> 16) & 1) + (((uint32_t)lane >> 1) & 7) + 1, (r5 & 1) + ((uint32_t)lane & 1)));
        lds_read32x4<W, W, W, W>(lds, at0, at1, at0 + down, at1 + down, a0, a1, b0, b1);
        if (r0 & kROhC)
            return rc_mc4_hv(a0, a1, b0, b1, r4, r4 >> 8, k.ones);
        return rc_mc4_v(a0, a1, b0, b1, r4, k.ones);
    }
    lds_read32x2<W, W>(lds, at0, at1, a0, a1);
    if (r0 & kROhC)
        return rc_mc4_h(a0, a1, r4, r4 >> 8, k.ones);
    return funnel32(a1, a0, r4);
}

// ---- a gathered window (kRSlow: rows lie linearly, 32 bytes of luma / 16 of chroma each, from the dword below the origin)
MPG_HD uint32_t rc_mc4_any(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t shift_bytes, uint32_t ones, bool oh, bool ov)
{
    const uint32_t sh = shift_bytes * 8, sh8 = sh + 8;
    if (ov)
        return oh ? rc_mc4_hv(a0, a1, b0, b1, sh, sh8, ones) : rc_mc4_v(a0, a1, b0, b1, sh, ones);
    return oh ? rc_mc4_h(a0, a1, sh, sh8, ones) : funnel32(a1, a0, sh);
}
// dword d (0..7 luma, 0..3 chroma) of linear row `piece0`'s pieces in the gathered layout: piece = piece0 + (d >> 2), dword d & 3
MPG_HD uint32_t rc_gathered_dword(const uint8_t *win, uint32_t piece0, uint32_t d)
{
    return *reinterpret_cast<const uint32_t *>(win + (d & 3) * kRcGatherStride + 4 * (piece0 + (d >> 2)));
}
MPG_HD uint32_t rc_mc_luma_slow(const uint8_t *win, int lane, uint32_t r0, uint32_t r3, uint32_t ones)
{
    const uint32_t row = (uint32_t)lane >> 2, q = (uint32_t)lane & 3; // luma row r's two pieces: 2 r, 2 r + 1
    return rc_mc4_any(rc_gathered_dword(win, 2 * row, q), rc_gathered_dword(win, 2 * row, q + 1), rc_gathered_dword(win, 2 * row + 2, q),
                      rc_gathered_dword(win, 2 * row + 2, q + 1), r3 & 3, ones, (r0 & kROhL) != 0, (r0 & kROvL) != 0);
}
MPG_HD uint32_t rc_mc_chroma_slow(const uint8_t *win, int lane, uint32_t r0, uint32_t r4, uint32_t ones)
{
    const uint32_t l = (uint32_t)lane, piece = 34 + ((l >> 4) & 1) * 9 + ((l >> 1) & 7), h = l & 1; // (plane, row) -> its one piece
    return rc_mc4_any(rc_gathered_dword(win, piece, h), rc_gathered_dword(win, piece, h + 1), rc_gathered_dword(win, piece + 1, h),
                      rc_gathered_dword(win, piece + 1, h + 1), r4 & 3, ones, (r0 & kROhC) != 0, (r0 & kROvC) != 0);
}

// ---- step 4: residual row + the 8 prediction bytes in O_m -> clamped bytes (video.go:943-971)
MPG_HD void rc_rmw(uint8_t *lds, uint32_t bw, int lane, const int32_t (&v)[8])
{
    uint8_t *p = lds + ((bw << 3) & 0xff8u) + ((uint32_t)lane & 7) * ((bw & kBChroma) ? 8u : 16u);
    const uint64_t pred = *reinterpret_cast<const uint64_t *>(p);
    *reinterpret_cast<uint64_t *>(p) = add_clamp_pack8(pred, v);
}

// ---- step 5: stores (tiled frame: a macroblock's luma is 256 contiguous bytes, its Cb and Cr 64 each)

// horizontal run = 4 consecutive tiles: luma 1 KB by all 64 lanes (16 bytes each), the four Cb | Cr pairs 512 bytes by lanes
// 0..31, as they lie in the O_m; where they go the header says (h6 / h7, from the wave's one frame base).  Non-temporal stores:
// the picture is next read by a later launch; the prediction windows of the neighbouring chunks, which ARE read again within
// microseconds, keep their place in L2 (profiles/r4z_ab_non_temporal_frame_stores.txt: typical +1.9 %, dense +0.3 %; the
// fused-RGBA instance likewise, with its RGBA stores: profiles/r5_ab_*: +3.8 % / +1.6 %)
MPG_HD void rc_store_run_luma(const VideoArgs &a, const RcChunk &c, int lane, const uint8_t *lds)
{
    const uint32_t l = (uint32_t)lane;
    store16_at<true>(rc_frame_base(a, c), c.h[6] + l * 16, *reinterpret_cast<const u32x4 *>(lds + rc_win_at(l >> 4) + (l & 15) * 16));
}
MPG_HD void rc_store_run_chroma(const VideoArgs &a, const RcChunk &c, int lane, const uint8_t *lds)
{
    const uint32_t l = (uint32_t)lane;
    if (lane < 32)
        store16_at<true>(rc_frame_base(a, c), c.h[7] + l * 16, *reinterpret_cast<const u32x4 *>(lds + rc_win_at(l >> 3) + 256 + (l & 7) * 16));
}
MPG_HD void rc_store_run(const VideoArgs &a, const RcChunk &c, int lane, const uint8_t *lds)
{
    rc_store_run_luma(a, c, lane, lds);
    rc_store_run_chroma(a, c, lane, lds);
}

// any other chunk: macroblock m by lanes (block b = lane>>3, row j = lane&7), 8 bytes each.  An invalid intra
// block keeps the frame's pixels (video.go:711-714); with `mirror` its bytes are parked in O_m instead, so that
// the colour conversion sees what the planes hold.
MPG_HD void rc_store_mb(const VideoArgs &a, const RcChunk &c, uint32_t m, int lane, uint8_t *lds, bool mirror)
{
    const int b = lane >> 3, j = lane & 7;
    if (b >= 6)
        return;
    const uint32_t d0 = c.r[m][0];
    const bool written = !(d0 & kRIntra) || ((d0 >> 8) & 0x3f & (0x20u >> b)) != 0;
    const uint32_t mb = rc_mb_index(a, d0);
    uint32_t off;
    if (b < 4)
        off = mb * 256 + ((uint32_t)j + ((uint32_t)(b >> 1) << 3)) * 16 + ((uint32_t)(b & 1) << 3);
    else
        off = a.luma_bytes + (uint32_t)(b - 4) * kChromaCrAt + mb * kChromaBlockStep + (uint32_t)j * 8;
    uint8_t *cur = rc_frame_base(a, c) + rc_cur_offset(a, c) + off;
    uint8_t *t = lds + rc_tile_offset(b, j, m);
    if (written)
        *reinterpret_cast<uint64_t *>(cur) = *reinterpret_cast<const uint64_t *>(t);
    else if (mirror)
        *reinterpret_cast<uint64_t *>(t) = *reinterpret_cast<const uint64_t *>(cur);
}

// The host mirror (mpeghip_video_host_mirror; recon_wide_kernel<false, true>): macroblock m's 384 output bytes once more, into the
// reference's LINEAR planes (video.go:347-355: Y | Cb | Cr, rows of luma_w / chroma_w bytes) of a frame in pinned host memory —
// lanes 0..15: luma row `lane` (16 bytes), 16..23: Cb row (8 bytes), 24..31: Cr row.  O_m must hold the macroblock as the frame
// store holds it: always so in a run; in any other chunk rc_store_mb has parked the pixels an invalid intra block keeps (mirror).
MPG_HD void rc_mirror_mb(const VideoArgs &a, const RcChunk &c, uint8_t *frame, uint32_t m, int lane, const uint8_t *lds)
{
    const uint32_t d0 = c.r[m][0], mb_x = (d0 >> 16) & 0xff, mb_y = d0 >> 24, l = (uint32_t)lane;
    const uint8_t *O = lds + rc_win_at(m);
    if (l < 16) {
        *reinterpret_cast<u32x4 *>(frame + (size_t)((mb_y << 4) + l) * a.luma_w + (mb_x << 4)) = *reinterpret_cast<const u32x4 *>(O + l * 16);
    } else if (l < 32) {
        const uint32_t plane = (l >> 3) & 1, r = l & 7;
        *reinterpret_cast<uint64_t *>(frame + a.luma_bytes + (size_t)plane * a.chroma_bytes + (size_t)((mb_y << 3) + r) * a.chroma_w + (mb_x << 3)) =
            *reinterpret_cast<const uint64_t *>(O + 256 + plane * 64 + r * 8);
    }
}

// Frame.RGBA fused (pictures flagged MPEGHIP_PIC_RGBA): macroblock m from O_m, 4 pixels per lane (lane = row*4 +
// segment), one 16-byte store each — a macroblock row is 64 contiguous bytes of the image.  Pixels outside
// width x height are not stored.
// (mb_x, mb_y: the macroblock's position — of a run's macroblock m it is the first one's + m, no record is indexed by lane)
MPG_HD uint8_t *rc_rgba_image(const VideoArgs &a, const RcChunk &c)
{
    return a.rgba + ((uint64_t)rc_stream(c) * MPEGHIP_SLOTS + rc_cur_slot(c)) * a.rgba_stride; // wave-uniform
}
MPG_HD void rc_rgba_quad(const VideoArgs &a, uint8_t *img, uint32_t m, uint32_t mb_x, uint32_t mb_y, uint32_t row, uint32_t seg, const uint8_t *lds)
{
    const uint32_t py = (mb_y << 4) + row, px0 = (mb_x << 4) + seg * 4;
    if (py >= a.height || px0 >= a.width)
        return;
    const uint8_t *O = lds + rc_win_at(m);
    const uint32_t yy = *reinterpret_cast<const uint32_t *>(O + row * 16 + seg * 4);
    const uint32_t cb = *reinterpret_cast<const uint16_t *>(O + 256 + (row >> 1) * 8 + seg * 2);
    const uint32_t cr = *reinterpret_cast<const uint16_t *>(O + 320 + (row >> 1) * 8 + seg * 2);
    uint32_t px[4];
    rgba_row4(yy, chroma_terms(cb & 0xff, cr & 0xff), chroma_terms((cb >> 8) & 0xff, (cr >> 8) & 0xff), px);
    const uint64_t p = (uint64_t)py * a.width + px0;
    const uint32_t n = a.width - px0 >= 4 ? 4 : a.width - px0;
    rgba_store4<true>(reinterpret_cast<uint32_t *>(img) + p, p, px, n); // (non-temporal: rc_store_run)
}
MPG_HD void rc_rgba_mb(const VideoArgs &a, const RcChunk &c, uint8_t *img, uint32_t m, int lane, const uint8_t *lds)
{
    const uint32_t d0 = c.r[m][0];
    rc_rgba_quad(a, img, m, (d0 >> 16) & 0xff, d0 >> 24, (uint32_t)lane >> 2, (uint32_t)lane & 3, lds);
}
// a horizontal run of 4 macroblocks, rows 4q .. 4q + 3: lane = (row lane>>4, macroblock (lane>>2)&3, segment lane&3), so that
// one store instruction writes 4 rows of 256 contiguous bytes (whole cache lines) instead of 16 rows of 64
MPG_HD void rc_rgba_run_rows(const VideoArgs &a, const RcChunk &c, uint8_t *img, uint32_t q, int lane, const uint8_t *lds)
{
    const uint32_t l = (uint32_t)lane, m = (l >> 2) & 3, d0 = c.r[0][0];
    rc_rgba_quad(a, img, m, ((d0 >> 16) & 0xff) + m, d0 >> 24, q * 4 + (l >> 4), l & 3, lds);
}

} // namespace mpg
