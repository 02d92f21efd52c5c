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

#include <type_traits>
#include <vector>

#include "audio_lane.h"
#include "video_lane.h"
#include "video_pack_lane.h"
#include "video_recon_lane.h"

using namespace mpg;

static const uint8_t kPremult[64] = {32, 44, 42, 38, 32, 25, 17, 9,  44, 62, 58, 52, 44, 35, 24, 12, 42, 58, 55, 49, 42, 33,
                                     23, 12, 38, 52, 49, 44, 38, 30, 20, 10, 32, 44, 42, 38, 32, 25, 17, 9,  25, 35, 33, 30,
                                     25, 20, 14, 7,  17, 24, 23, 20, 17, 14, 9,  5,  9,  12, 12, 10, 9,  7,  5,  2};

template <bool kWide>
static uint32_t emu_pack_as(uint32_t luma_w, uint32_t luma_h, uint64_t frame_stride, uint64_t rgba_stride, const mpeghip_pic_desc *pic,
                            const mpeghip_mb_desc *mbs, const uint8_t *coefs, uint32_t *chunks_out, uint32_t *words_out, uint32_t *n_words)
{
    RcGeom geom;
    geom.mb_w = luma_w / 16;
    geom.mb_h = luma_h / 16;
    geom.luma_w = luma_w;
    geom.chroma_w = luma_w / 2;
    geom.luma_bytes = luma_w * luma_h;
    geom.frame_stride = frame_stride;
    geom.rgba_stride = rgba_stride;
    const RcPacked got = rc_pack_picture<kWide>(geom, *pic, mbs, pic->mb_count, coefs, 0, chunks_out, words_out);
    *n_words = got.words;
    return got.chunks;
}
static thread_local int g_device_pack = 0; // (per thread: ShardedVideoBatch tests run two emulator stores on two threads)
// 1: sparse pictures are packed by the DEVICE packer's lane functions (video_pack_lane.h), wave by wave
static uint32_t g_pack_window = kPkWinDwords; // the device packer's LDS window (pack_kernel: kPkWinDwords)
static int g_tile_policy = 0; // as mpeghip_video_set_tile_policy: 0 = pick per submit like launch_batch, 1 = the DPP instance, 2 = the instance for dense units
extern "C" {

void emu_set_tile_policy(int policy) { g_tile_policy = policy; }
static int g_wide = 0; // 1: every chunk runs as recon_wide_kernel runs it — four waves, two barriers (emu_wide_chunk)
static uint64_t g_wide_chunks = 0; // chunks run that way so far (tests ask: did the switch take?)
void emu_set_wide(int on) { g_wide = on; }
int emu_get_wide(void) { return g_wide; }
// the host mirror (mpeghip_video_host_mirror): where emu_wide_chunk writes every macroblock once more, linearly — recon_wide_kernel<false,
// true>'s rc_mirror_mb — for submits without a colour-converting picture; (nullptr, 0) = off
static uint8_t *g_mirror = nullptr;
static uint64_t g_mirror_stride = 0;
void emu_set_mirror(uint8_t *base, uint64_t stride) { g_mirror = base, g_mirror_stride = stride; }
uint64_t emu_wide_chunks_run(void) { return g_wide_chunks; }
void emu_set_device_pack(int on) { g_device_pack = on; }
void emu_set_pack_window(uint32_t dwords) { g_pack_window = dwords; }

// pack_kernel for ONE picture, wave by wave: its descriptors and sparse words (n_words dwords behind words_in + aux.word_first)
// -> chunks (from aux.chunk_first) and words_out (from aux.word_first).  Returns the error word (kPkNoError = fine).
unsigned long long emu_pack_device_picture(uint32_t luma_w, uint32_t luma_h, uint64_t frame_stride, uint64_t rgba_stride,
                                           const mpeghip_pic_desc *pic, uint32_t word_first, uint32_t n_words, uint32_t chunk_first,
                                           const mpeghip_mb_desc *mbs, const uint32_t *words_in, uint32_t *chunks_out, uint32_t *words_out,
                                           uint32_t *use_out)
{
    PackArgs a;
    mpeghip_pic_desc p = *pic;
    PkPic aux{word_first, n_words, chunk_first, 0};
    unsigned long long err = kPkNoError;
    a.pics = &p;
    a.aux = &aux;
    a.mbs = mbs;
    a.words_in = words_in;
    a.chunks = chunks_out;
    a.words_out = words_out;
    a.mb_w = luma_w / 16;
    a.mb_h = luma_h / 16;
    a.seen_stride = (a.mb_w * a.mb_h + 31) / 32;
    std::vector<uint32_t> seen(a.seen_stride, 0);
    a.seen = seen.data();
    a.err = &err;
    a.n_pics = 1;
    a.win_dwords = g_pack_window;
    a.groups_per_pic = (p.mb_count + 63) / 64;
    a.luma_w = luma_w;
    a.chroma_w = luma_w / 2;
    a.luma_bytes = luma_w * luma_h;
    a.chroma_bytes = a.luma_bytes / 4;
    a.frame_bytes = (uint64_t)a.luma_bytes + 2 * a.chroma_bytes + (uint64_t)luma_w * 16;
    a.frame_stride = frame_stride;
    a.rgba_stride = rgba_stride;
    uint32_t use = 0;
    for (uint32_t g = 0; g < a.groups_per_pic; g++) {
        uint32_t xch[64 * kPkXchDwords];
        PkLane L[64];
        const uint32_t k_next = g * 64 + 64;
        const uint32_t next_coef_off = k_next < p.mb_count ? mbs[p.mb_first + k_next].coef_off : 0u;
        // phase 0: the wave's window (g_pack_window dwords at most: tests shrink it so that reads fall on both sides of its end)
        std::vector<uint32_t> win(g_pack_window + 4, 0xCDCDCDCDu);
        PkWin in;
        in.glob = words_in + word_first;
        in.lds = win.data();
        pk_window_range(mbs[p.mb_first + g * 64].coef_off, k_next < p.mb_count ? next_coef_off : n_words, n_words, g_pack_window, in.lo, in.n);
        for (uint32_t i = 0; i < in.n; i += 4) // (16 bytes per lane and load: up to 3 dwords past the range, inside the padded buffer)
            memcpy(win.data() + i, in.glob + in.lo + i, 16);
        std::vector<uint32_t> wout(g_pack_window + 4, 0xEFEFEFEFu); // the window of the produced words (uninitialised LDS on the device)
        PkOut out;
        out.glob = words_out;
        out.lds = wout.data();
        out.base = word_first + in.lo;
        out.n = in.n;
        for (int lane = 0; lane < 64; lane++) {
            const uint32_t k = g * 64 + (uint32_t)lane;
            mpeghip_mb_desc mb;
            memset(&mb, 0, sizeof(mb));
            if (k < p.mb_count)
                mb = mbs[p.mb_first + k];
            L[lane] = pk_scan(a, 0, p, aux, k, mb, in);
            use |= L[lane].ok ? L[lane].use : 0u;
        }
        for (int lane = 0; lane < 64; lane++)
            pk_share(xch, lane, L[lane]);
        for (int lane = 0; lane < 64; lane++)
            pk_emit(a, p, aux, g * 64 + (uint32_t)lane, lane, L[lane], xch, next_coef_off, in, out);
        for (uint32_t i = mbs[p.mb_first + g * 64].coef_off - in.lo; i < out.n; i++) // (the wave's own territory only)
            words_out[out.base + i] = wout[i];
    }
    if (use_out)
        *use_out = use;
    return err;
}

// one stream's dequantisation table in the device layout (what mpeghip_video_open / _set_quant upload)
void emu_make_qtable(uint8_t *out, const uint8_t *intra, const uint8_t *non_intra) { rc_make_qtable(out, intra, non_intra, kPremult); }

// recon_kernel, wave by wave: the library's packer (rc_pack_picture) turns the ABI arrays into the device
// format, then every chunk runs through the kernel's lane functions in the kernel's order, LDS = an array.
static int emu_video_run_form(uint8_t *frames, uint64_t frame_stride, uint32_t luma_w, uint32_t luma_h,
                  uint32_t width, uint32_t height,
                  const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                  const uint8_t *coefs, const uint8_t *qtable, uint8_t *rgba, uint64_t rgba_stride, uint64_t sparse_words);
// the unit form (mpeghip_video_submit)
int emu_video_run(uint8_t *frames, uint64_t frame_stride, uint32_t luma_w, uint32_t luma_h, uint32_t width, uint32_t height,
                  const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                  const uint8_t *coefs, const uint8_t *qtable, uint8_t *rgba, uint64_t rgba_stride)
{
    return emu_video_run_form(frames, frame_stride, luma_w, luma_h, width, height, pics, n_pics, mbs, n_mbs, coefs, qtable, rgba,
                              rgba_stride, 0);
}
// the sparse hand-over (mpeghip_video_stage_put_sparse): `words` checked by the packer as in the library; -2 = malformed
int emu_video_run_sparse(uint8_t *frames, uint64_t frame_stride, uint32_t luma_w, uint32_t luma_h, uint32_t width, uint32_t height,
                         const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                         const uint32_t *words, uint64_t n_words, const uint8_t *qtable, uint8_t *rgba, uint64_t rgba_stride)
{
    return emu_video_run_form(frames, frame_stride, luma_w, luma_h, width, height, pics, n_pics, mbs, n_mbs,
                              reinterpret_cast<const uint8_t *>(words), qtable, rgba, rgba_stride, n_words + 1);
}
// recon_wide_kernel (mpeghip.hip), one chunk: FOUR waves.  Wave w fetches the table and window w, runs residual pass w (its own
// int16 tile behind the four windows) and the motion compensation of macroblock w; a workgroup barrier; every wave adds its
// residual rows to the output bytes (whichever macroblock's they are); a barrier; the stores, every wave its share (a run: luma by
// wave 0, chroma by wave 1, four image rows of the colour conversion each — macroblock w if the run wraps a row end; any other
// chunk: macroblock w).  Between two barriers the waves run concurrently on the device and touch disjoint LDS (the table: the same
// bytes): here one after the other, in the kernel's statement order.  Poisoned LDS, as in the one-wave emulation.
static void emu_wide_chunk(const VideoArgs &a, uint32_t chunk, bool any_rgba)
{
    alignas(16) uint8_t lds[kRcTileAt + 3 * kRcTileBytes16];
    memset(lds, 0xCD, sizeof(lds));
    const RcChunk c = rc_load_chunk(a, chunk);
    const uint32_t n_blocks = rc_n_blocks(c);
    RcLane k[64];
    for (int lane = 0; lane < 64; lane++)
        k[lane] = rc_lane(a, lane);
    int32_t v[4][64][8];
    uint32_t bw[4][64];
    const int win_at[4] = {kRcWinAt, kRcWinAt + kRcWinBytes, kRcWinAt + 2 * kRcWinBytes, kRcWinAt + 3 * kRcWinBytes};
    // ---- up to the first barrier: wave w's loads, pass and motion compensation
    for (uint32_t w = 0; w < 4; w++) {
        const bool my_pass = w * 8 < n_blocks;
        uint32_t ent_at = 0, np = 0;
        for (uint32_t p = 0; p < w && p < 3; p++)
            ent_at += rc_pass_entries(c, p);
        if (w < 3)
            np = rc_pass_entries(c, w);
        for (int lane = 0; lane < kRcQtabBytes / kRcPiece; lane++) // (its own 12 lanes only: another wave's window is not written over)
            memcpy(lds + kRcQtabAt + 16 * lane, a.qmat + rc_table_lane_offset(c, lane), 16);
        for (int lane = 0; lane < kRcWinLanes; lane++) {
            const uint32_t off[4] = {rc_win_offset(c, 0, k[lane]), rc_win_offset(c, 1, k[lane]), rc_win_offset(c, 2, k[lane]), rc_win_offset(c, 3, k[lane])};
            memcpy(lds + win_at[w] + 16 * lane, rc_frame_base(a, c) + off[w] + win_at[w], 16);
        }
        if (c.r[w][0] & kRSlow) // a window that leaves its plane: gathered behind the regular load, into its place
            for (int lane = 0; lane < kRcGatherLanes; lane++)
                switch (w) {
                case 0: rc_gather_to_lds<0>(a, c, rc_frame_base(a, c), lds, lane); break;
                case 1: rc_gather_to_lds<1>(a, c, rc_frame_base(a, c), lds, lane); break;
                case 2: rc_gather_to_lds<2>(a, c, rc_frame_base(a, c), lds, lane); break;
                default: rc_gather_to_lds<3>(a, c, rc_frame_base(a, c), lds, lane); break;
                }
        if (my_pass) {
            int16_t *T16 = reinterpret_cast<int16_t *>(lds + kRcTileAt + w * kRcTileBytes16);
            uint32_t e[64];
            for (int lane = 0; lane < 64; lane++) {
                e[lane] = load32_uncounted(rc_word_base(a, c), rc_ent_lane_offset(c, ent_at, lane));
                bw[w][lane] = load32_uncounted(rc_word_base(a, c), rc_blk_lane_offset(w, lane));
            }
            if (np) {
                for (int lane = 0; lane < 64; lane++)
                    rc_zero_tile16(T16, lane);
                for (uint32_t r = 0; r < np; r += 64)
                    for (int lane = 0; lane < 64; lane++) {
                        if (r > 0)
                            e[lane] = *rc_ent_src(a, c, ent_at + r, lane);
                        if (r + (uint32_t)lane < np)
                            rc_scatter16(T16, lds, e[lane]);
                    }
            }
            for (int lane = 0; lane < 64; lane++) {
                const bool mine = w * 8 + ((uint32_t)lane >> 3) < n_blocks;
                if (np)
                    rc_cols_load16(T16, lds, lane, v[w][lane]);
                else
                    for (int r = 0; r < 8; r++)
                        v[w][lane][r] = 0;
                if (rc_any_special(c)) {
                    if (rc_any_dcword(c) && mine)
                        rc_dc_from_word(bw[w][lane], lane, v[w][lane]);
                    if (rc_any_raw(c) && mine && (bw[w][lane] & kBRaw))
                        rc_raw_cols(a, c, bw[w][lane], lane, v[w][lane]);
                    if (rc_any_dense(c) && mine && (bw[w][lane] & kBDense))
                        rc_dense_cols<false>(rc_dense_read(a, c, bw[w][lane], lane), lds, bw[w][lane], lane, v[w][lane]);
                }
                idct8<false>(v[w][lane]);
            }
            for (int g = 0; g < 8; g++) { // the transposition across the block's 8 lanes: lane j leaves with row j
                int32_t m[8][8];
                for (int j = 0; j < 8; j++)
                    for (int r = 0; r < 8; r++)
                        m[r][j] = v[w][g * 8 + j][r];
                for (int j = 0; j < 8; j++)
                    for (int col = 0; col < 8; col++)
                        v[w][g * 8 + j][col] = m[j][col];
            }
            for (int lane = 0; lane < 64; lane++)
                idct8<true>(v[w][lane]);
        }
        // motion compensation of macroblock w
        const uint32_t r0 = c.r[w][0];
        if (r0 & kRDead)
            continue;
        uint8_t *win = lds + rc_win_at(w);
        uint32_t yl[64], yc[64];
        for (int lane = 0; lane < 64; lane++) {
            yl[lane] = yc[lane] = 0;
            if (!(r0 & (kRIntra | kRSlow))) {
                switch (w) { // (the kernel's by_wave: the macroblock is a template argument)
                case 0: yl[lane] = rc_mc_luma<0>(lds, k[lane], r0, c.r[0][3]), yc[lane] = rc_mc_chroma<0>(lds, k[lane], lane, r0, c.r[0][4], c.r[0][5]); break;
                case 1: yl[lane] = rc_mc_luma<1>(lds, k[lane], r0, c.r[1][3]), yc[lane] = rc_mc_chroma<1>(lds, k[lane], lane, r0, c.r[1][4], c.r[1][5]); break;
                case 2: yl[lane] = rc_mc_luma<2>(lds, k[lane], r0, c.r[2][3]), yc[lane] = rc_mc_chroma<2>(lds, k[lane], lane, r0, c.r[2][4], c.r[2][5]); break;
                default: yl[lane] = rc_mc_luma<3>(lds, k[lane], r0, c.r[3][3]), yc[lane] = rc_mc_chroma<3>(lds, k[lane], lane, r0, c.r[3][4], c.r[3][5]); break;
                }
            } else if (r0 & kRSlow) {
                yl[lane] = rc_mc_luma_slow(win, lane, r0, c.r[w][3], k[lane].ones);
                yc[lane] = rc_mc_chroma_slow(win, lane, r0, c.r[w][4], k[lane].ones);
            }
        }
        for (int lane = 0; lane < 64; lane++) { // (over the window: only after every lane has its taps)
            memcpy(win + k[lane].out_luma, &yl[lane], 4);
            memcpy(win + k[lane].out_chroma, &yc[lane], 4);
        }
    }
    // ---- barrier: the four O_m are complete; every wave's residual rows onto them
    for (uint32_t w = 0; w < 4; w++)
        if (w * 8 < n_blocks)
            for (int lane = 0; lane < 64; lane++)
                if (w * 8 + ((uint32_t)lane >> 3) < n_blocks)
                    rc_rmw(lds, bw[w][lane], lane, v[w][lane]);
    // ---- barrier: the stores, every wave its share
    const bool run = (c.h[4] & kCRun) != 0, to_rgba = any_rgba && (c.h[4] & kCRgba) != 0;
    const bool mirror = g_mirror && !any_rgba; // (launch_batch: the mirroring instance carries no colour conversion)
    uint8_t *const mirror_frame = mirror ? g_mirror + ((uint64_t)rc_stream(c) * MPEGHIP_SLOTS + rc_cur_slot(c)) * g_mirror_stride : nullptr;
    const uint32_t n_live = rc_n_live(c);
    for (uint32_t w = 0; w < 4; w++) {
        if (run) {
            for (int lane = 0; lane < 64; lane++) {
                if (w == 0)
                    rc_store_run_luma(a, c, lane, lds);
                if (w == 1)
                    rc_store_run_chroma(a, c, lane, lds);
                if (to_rgba) {
                    if (rc_run_in_one_row(c))
                        rc_rgba_run_rows(a, c, rc_rgba_image(a, c), w, lane, lds);
                    else
                        rc_rgba_mb(a, c, rc_rgba_image(a, c), w, lane, lds);
                }
                if (mirror)
                    rc_mirror_mb(a, c, mirror_frame, w, lane, lds);
            }
        } else if (w < n_live) {
            for (int lane = 0; lane < 64; lane++)
                rc_store_mb(a, c, w, lane, lds, to_rgba || mirror);
            if (to_rgba)
                for (int lane = 0; lane < 64; lane++)
                    rc_rgba_mb(a, c, rc_rgba_image(a, c), w, lane, lds);
            if (mirror)
                for (int lane = 0; lane < 64; lane++)
                    rc_mirror_mb(a, c, mirror_frame, w, lane, lds);
        }
    }
}

static int emu_video_run_form(uint8_t *frames, uint64_t frame_stride, uint32_t luma_w, uint32_t luma_h,
                  uint32_t width, uint32_t height,
                  const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                  const uint8_t *coefs, const uint8_t *qtable, uint8_t *rgba, uint64_t rgba_stride, uint64_t sparse_words)
{
    VideoArgs a;
    a.frames = frames;
    a.frames_b = frames - kRcDmaBias;
    a.frame_stride = frame_stride;
    a.mb_w = luma_w / 16;
    a.mb_h = luma_h / 16;
    a.luma_w = luma_w;
    a.luma_h = luma_h;
    a.chroma_w = luma_w / 2;
    a.chroma_h = luma_h / 2;
    a.luma_bytes = luma_w * luma_h;
    a.chroma_bytes = a.luma_bytes / 4;
    if (frame_stride % 256 || rgba_stride % 256)
        abort(); // the chunks name frames in units of 256 bytes
    RcGeom geom;
    geom.mb_w = a.mb_w;
    geom.mb_h = a.mb_h;
    geom.luma_w = a.luma_w;
    geom.chroma_w = a.chroma_w;
    geom.luma_bytes = a.luma_bytes;
    geom.frame_stride = frame_stride;
    geom.rgba_stride = rgba_stride;
    bool any_rgba = false; // the product picks the kernel instance by this (mpeghip.hip: launch_batch)
    uint64_t units = 0, n_chunks = 0;
    for (uint32_t p = 0; p < n_pics; p++) {
        any_rgba = any_rgba || (pics[p].flags & MPEGHIP_PIC_RGBA);
        n_chunks += rc_max_chunks(pics[p].mb_count);
    }
    // which pictures are in the sparse form: all of them (emu_video_run_sparse), or those that say so (MPEGHIP_PIC_SPARSE)
    std::vector<uint8_t> pic_sparse(n_pics, 0);
    uint64_t sparse_dwords = sparse_words ? sparse_words - 1 : 0;
    for (uint32_t p = 0; p < n_pics; p++) {
        pic_sparse[p] = sparse_words || (pics[p].flags & MPEGHIP_PIC_SPARSE);
        for (uint32_t i = pics[p].mb_first; i < pics[p].mb_first + pics[p].mb_count; i++) {
            const bool raw = (mbs[i].flags & MPEGHIP_MB_COEF_RAW) != 0;
            const uint32_t nb = (uint32_t)__builtin_popcount(mbs[i].cbp & 0x3f);
            if (!pic_sparse[p]) {
                const uint64_t end = (uint64_t)mbs[i].coef_off + (uint64_t)nb * (raw ? 2 : 1);
                units = nb && end > units ? end : units;
            } else if (!sparse_words) { // (the host parser's own words: their extent by walking them)
                uint64_t at = mbs[i].coef_off;
                for (uint32_t b = 0; b < nb; b++)
                    at += 1 + (raw ? 64 : reinterpret_cast<const uint32_t *>(coefs)[at]);
                sparse_dwords = at > sparse_dwords ? at : sparse_dwords;
            }
        }
    }
    std::vector<uint32_t> chunks(n_chunks * kRcChunkDwords + 4),
        words((g_device_pack ? (size_t)n_pics : 1) * (rc_max_words_sparse(sparse_dwords, n_mbs) + 16) + rc_max_words(units) + kRcWordsPad, 0xDEADBEEFu);
    uint32_t nc = 0, nw = 0;
    uint64_t coded = 0, dense = 0;
    std::vector<uint32_t> staged; // (device packer: the picture's words where the staged copy would have them)
    for (uint32_t p = 0; p < n_pics; p++) {
        if (pic_sparse[p] && g_device_pack) {
            const uint32_t n_sparse = (uint32_t)(sparse_words ? sparse_words - 1 : sparse_dwords);
            staged.assign(words.size(), 0xDEADBEEFu);
            if (n_sparse)
                memcpy(staged.data() + nw, coefs, (size_t)n_sparse * 4);
            mpeghip_pic_desc pd = pics[p];
            const uint32_t pic_chunks = (uint32_t)rc_max_chunks(pd.mb_count);
            if (emu_pack_device_picture(luma_w, luma_h, frame_stride, rgba_stride, &pd, nw, n_sparse, nc, mbs, staged.data(), chunks.data(),
                                        words.data(), nullptr) != kPkNoError)
                return -2;
            for (uint32_t c = nc; c < nc + pic_chunks; c++) { // which kernel instance suits the batch: as the host packer counts
                const uint32_t *h = chunks.data() + (size_t)c * kRcChunkDwords;
                coded += h[5] & kHBlocksMask;
                for (uint32_t i = 0; i < (h[5] & kHBlocksMask); i++)
                    dense += (words[rc_chunk_word_index(h) + i] & kBDense) ? 1 : 0;
            }
            nc += pic_chunks;
            nw += (n_sparse + 15) / 16 * 16; // (as the product's staging: a picture's words begin on a 64-byte boundary)
            continue;
        }
        const RcPacked got = pic_sparse[p] ? rc_pack_picture<true, true>(geom, pics[p], mbs + pics[p].mb_first, pics[p].mb_count, coefs, nw,
                                                                        chunks.data() + (size_t)nc * kRcChunkDwords, words.data() + nw,
                                                                        sparse_words ? sparse_words - 1 : ~0ull >> 2)
                                          : rc_pack_picture(geom, pics[p], mbs + pics[p].mb_first, pics[p].mb_count, coefs, nw,
                                                            chunks.data() + (size_t)nc * kRcChunkDwords, words.data() + nw);
        if (got.bad)
            return -2;
        nc += got.chunks;
        nw += got.words;
        coded += got.blocks;
        dense += got.dense_blocks;
    }
    // which kernel instance: the product's rule (mpeghip.hip: launch_batch), unless a test pins one
    bool t16 = dense * 8 <= coded; // (kDenseShareNum / kDenseShareDen)
    if (g_tile_policy)
        t16 = g_tile_policy == 1;
    a.pics = pics;
    a.chunks = chunks.data();
    a.words = words.data();
    a.qmat = qtable;
    a.n_chunks = nc;
    a.width = width;
    a.height = height;
    a.rgba = rgba;
    a.rgba_stride = rgba_stride;

    alignas(16) uint8_t lds[kRcLdsBytesMax]; // (no statics: ShardedVideoBatch tests run two emulator stores on two threads)
    for (uint32_t chunk = 0; chunk < nc; chunk++) {
        if (g_wide) { // recon_wide_kernel's orchestration of the same lane functions over the same chunk
            emu_wide_chunk(a, chunk, any_rgba);
            g_wide_chunks++;
            continue;
        }
        memset(lds, 0xCD, sizeof(lds)); // poison: reads of unwritten LDS must not matter
        int16_t *T16 = reinterpret_cast<int16_t *>(lds + kRcTileAt);
        const RcChunk c = rc_load_chunk(a, chunk);
        const uint32_t n_blocks = rc_n_blocks(c);
        RcLane k[64];
        uint32_t bw[64], e[64];
        for (int lane = 0; lane < 64; lane++) {
            k[lane] = rc_lane(a, lane);
            e[lane] = load32_uncounted(rc_word_base(a, c), rc_ent_lane_offset(c, 0, lane));
        }
        for (int i = 0; i < 5; i++) // (load by load, as they complete on the device: later loads overwrite the surplus lanes)
            for (int lane = 0; lane < kRcWinLanes; lane++) {
                const uint32_t off[5] = {rc_table_lane_offset(c, lane), rc_win_offset(c, 0, k[lane]), rc_win_offset(c, 1, k[lane]),
                                         rc_win_offset(c, 2, k[lane]), rc_win_offset(c, 3, k[lane])};
                const int at[5] = {kRcQtabAt, kRcWinAt, kRcWinAt + kRcWinBytes, kRcWinAt + 2 * kRcWinBytes, kRcWinAt + 3 * kRcWinBytes};
                // (the device adds the instruction's offset field — the LDS target — to the global address too: dma_table_and_windows)
                const uint8_t *src = i == 0 ? a.qmat + off[0] : rc_frame_base(a, c) + off[i] + at[i];
                memcpy(lds + at[i] + 16 * lane, src, 16);
            }
        // ... and, behind them, the gather of every window that leaves its plane (four one-dword loads by lanes 0..51 each)
        for (int lane = 0; lane < kRcGatherLanes; lane++) {
            if (c.r[0][0] & kRSlow)
                rc_gather_to_lds<0>(a, c, rc_frame_base(a, c), lds, lane);
            if (c.r[1][0] & kRSlow)
                rc_gather_to_lds<1>(a, c, rc_frame_base(a, c), lds, lane);
            if (c.r[2][0] & kRSlow)
                rc_gather_to_lds<2>(a, c, rc_frame_base(a, c), lds, lane);
            if (c.r[3][0] & kRSlow)
                rc_gather_to_lds<3>(a, c, rc_frame_base(a, c), lds, lane);
        }
        int32_t v[64][8];
        uint32_t ent_at = 0;
        bool table_flat = !t16 && rc_any_dense(c); // (as the kernel — the instance for dense units: the wave's AND over its lanes' columns)
        for (int lane = 0; lane < 64 && table_flat; lane++)
            table_flat = rc_non_intra_column_flat(lds, lane);
        auto residual_pass = [&](uint32_t pass) {
            const uint32_t np = rc_pass_entries(c, pass);
            for (int lane = 0; lane < 64; lane++)
                bw[lane] = rc_word_base(a, c)[rc_blk_lane_offset(pass, lane) / 4];
            bool flat = table_flat; // the short dequantisation: no intra unit among the pass's dense ones
            for (int lane = 0; lane < 64; lane++)
                if (pass * 8 + ((uint32_t)lane >> 3) < n_blocks && (bw[lane] & kBDense) && !(bw[lane] >> 31))
                    flat = false;
            auto dense_cols = [&](int lane) {
                const i32x4_a4 lv = rc_dense_read(a, c, bw[lane], lane);
                if (flat)
                    rc_dense_cols<true>(lv, lds, bw[lane], lane, v[lane]);
                else
                    rc_dense_cols<false>(lv, lds, bw[lane], lane, v[lane]);
            };
            if (np) {
                for (int lane = 0; lane < 64; lane++)
                    rc_zero_tile16(T16, lane);
                for (uint32_t r = 0; r < np; r += 64)
                    for (int lane = 0; lane < 64; lane++) {
                        if (pass > 0 || r > 0)
                            e[lane] = *rc_ent_src(a, c, ent_at + r, lane);
                        if (r + (uint32_t)lane < np)
                            rc_scatter16(T16, lds, e[lane]);
                    }
                ent_at += np;
            }
            for (int lane = 0; lane < 64; lane++) {
                const bool mine = pass * 8 + ((uint32_t)lane >> 3) < n_blocks;
                if (np)
                    rc_cols_load16(T16, lds, lane, v[lane]);
                else
                    for (int r = 0; r < 8; r++)
                        v[lane][r] = 0;
                if (rc_any_dcword(c) && mine)
                    rc_dc_from_word(bw[lane], lane, v[lane]);
                if (rc_any_raw(c) && mine && (bw[lane] & kBRaw))
                    rc_raw_cols(a, c, bw[lane], lane, v[lane]);
                if (rc_any_dense(c) && mine && (bw[lane] & kBDense))
                    dense_cols(lane);
                idct8<false>(v[lane]);
            }
            if (t16) {
                for (int g = 0; g < 8; g++) { // the kernel's transposition across the block's 8 lanes: lane j leaves with row j
                    int32_t m[8][8];
                    for (int j = 0; j < 8; j++)
                        for (int r = 0; r < 8; r++)
                            m[r][j] = v[g * 8 + j][r];
                    for (int j = 0; j < 8; j++)
                        for (int col = 0; col < 8; col++)
                            v[g * 8 + j][col] = m[j][col];
                }
            } else { // the kernel's transposition through LDS (rc_transpose8_lds): lanes 0..31, then lanes 32..63, over the dead tile
                int32_t *T = reinterpret_cast<int32_t *>(T16);
                for (int h = 0; h < 2; h++) {
                    for (int lane = 32 * h; lane < 32 * h + 32; lane++)
                        rc_tpose_store(T, lane, v[lane]);
                    for (int lane = 32 * h; lane < 32 * h + 32; lane++)
                        rc_tpose_load(T, lane, v[lane]);
                }
            }
            for (int lane = 0; lane < 64; lane++)
                idct8<true>(v[lane]);
        };
        auto add_residual = [&](uint32_t pass) {
            for (int lane = 0; lane < 64; lane++)
                if (pass * 8 + ((uint32_t)lane >> 3) < n_blocks)
                    rc_rmw(lds, bw[lane], lane, v[lane]);
        };
        if (n_blocks)
            residual_pass(0);
        auto mc_one = [&](auto M) { // (as the kernel: the common kind takes the record's scalars as they are)
            constexpr int m = decltype(M)::value;
            const uint32_t r0 = c.r[m][0];
            if (r0 & kRDead)
                return;
            uint8_t *win = lds + rc_win_at(m);
            uint32_t yl[64], yc[64];
            const bool fast = !(r0 & (kRIntra | kRDead | kRSlow));
            for (int lane = 0; lane < 64; lane++) {
                yl[lane] = yc[lane] = 0;
                if (fast) {
                    yl[lane] = rc_mc_luma<m>(lds, k[lane], r0, c.r[m][3]);
                    yc[lane] = rc_mc_chroma<m>(lds, k[lane], lane, r0, c.r[m][4], c.r[m][5]);
                } else if (r0 & kRSlow) {
                    yl[lane] = rc_mc_luma_slow(win, lane, r0, c.r[m][3], k[lane].ones);
                    yc[lane] = rc_mc_chroma_slow(win, lane, r0, c.r[m][4], k[lane].ones);
                }
            }
            for (int lane = 0; lane < 32; lane++) // (lanes 32..63 repeat lanes 0..31's chroma: the same bytes to the same place)
                if (yc[lane] != yc[lane + 32] || k[lane].out_chroma != k[lane + 32].out_chroma)
                    abort();
            for (int lane = 0; lane < 64; lane++) { // (over the window: only after every lane has its taps)
                memcpy(win + k[lane].out_luma, &yl[lane], 4);
                memcpy(win + k[lane].out_chroma, &yc[lane], 4);
            }
        };
        mc_one(std::integral_constant<int, 0>{});
        mc_one(std::integral_constant<int, 1>{});
        mc_one(std::integral_constant<int, 2>{});
        mc_one(std::integral_constant<int, 3>{});
        if (n_blocks) {
            add_residual(0);
            for (uint32_t pass = 1; pass * 8 < n_blocks; pass++) {
                residual_pass(pass);
                add_residual(pass);
            }
        }
        const bool run = (c.h[4] & kCRun) != 0, rgba_on = any_rgba && (c.h[4] & kCRgba) != 0;
        const uint32_t n_live = rc_n_live(c);
        if (run) {
            for (int lane = 0; lane < 64; lane++)
                rc_store_run(a, c, lane, lds);
        } else {
            for (uint32_t m = 0; m < n_live; m++)
                for (int lane = 0; lane < 64; lane++)
                    rc_store_mb(a, c, m, lane, lds, rgba_on);
        }
        if (rgba_on) {
            uint8_t *img = rc_rgba_image(a, c);
            if (run && rc_run_in_one_row(c)) {
                for (uint32_t q = 0; q < 4; q++)
                    for (int lane = 0; lane < 64; lane++)
                        rc_rgba_run_rows(a, c, img, q, lane, lds);
            } else { // (also a run that wraps a row end)
                for (uint32_t m = 0; m < n_live; m++)
                    for (int lane = 0; lane < 64; lane++)
                        rc_rgba_mb(a, c, img, m, lane, lds);
            }
        }
    }
    return 0;
}

// the packer alone, for tests that look at the device format: returns chunks, *n_words
uint32_t emu_pack(uint32_t luma_w, uint32_t luma_h, uint64_t frame_stride, uint64_t rgba_stride, const mpeghip_pic_desc *pic,
                  const mpeghip_mb_desc *mbs, const uint8_t *coefs, uint32_t *chunks_out, uint32_t *words_out, uint32_t *n_words)
{
    return emu_pack_as<true>(luma_w, luma_h, frame_stride, rgba_stride, pic, mbs, coefs, chunks_out, words_out, n_words);
}
// the packer without its 512-bit forms (what a CPU without AVX-512 runs): must write the same words
uint32_t emu_pack_narrow(uint32_t luma_w, uint32_t luma_h, uint64_t frame_stride, uint64_t rgba_stride, const mpeghip_pic_desc *pic,
                         const mpeghip_mb_desc *mbs, const uint8_t *coefs, uint32_t *chunks_out, uint32_t *words_out, uint32_t *n_words)
{
    return emu_pack_as<false>(luma_w, luma_h, frame_stride, rgba_stride, pic, mbs, coefs, chunks_out, words_out, n_words);
}
// the HOST packer on a picture in the sparse form (rc_pack_picture<., true>): -> chunks, 0xffffffff if it refuses the picture
uint32_t emu_pack_sparse(uint32_t luma_w, uint32_t luma_h, uint64_t frame_stride, uint64_t rgba_stride, const mpeghip_pic_desc *pic,
                         const mpeghip_mb_desc *mbs, const uint32_t *words, uint64_t n_words, uint32_t *chunks_out, uint32_t *words_out,
                         uint32_t *n_words_out, uint64_t out_room)
{
    RcGeom geom;
    geom.mb_w = luma_w / 16;
    geom.mb_h = luma_h / 16;
    geom.luma_w = luma_w;
    geom.chroma_w = luma_w / 2;
    geom.luma_bytes = luma_w * luma_h;
    geom.frame_stride = frame_stride;
    geom.rgba_stride = rgba_stride;
    const RcPacked got = rc_pack_picture<true, true>(geom, *pic, mbs + pic->mb_first, pic->mb_count, reinterpret_cast<const uint8_t *>(words), 0,
                                                     chunks_out, words_out, n_words, out_room);
    *n_words_out = got.words;
    return got.bad ? 0xffffffffu : got.chunks;
}

int emu_host_has_avx512(void)
{
#if MPG_HOST_AVX512
    return rc_host_has_avx512() ? 1 : 0;
#else
    return 0;
#endif
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
            rgba_convert_quad(frame, luma_w / 16, luma_w * luma_h, luma_w * luma_h / 4, width, height, x4, y, rgba);
}

// the reference's linear planes (Y | Cb | Cr, n = luma_bytes + 2 chroma_bytes) <-> a slot of the tiled frame store
// (what mpeghip_video_read_planes / write_planes do with relayout_kernel)
void emu_relayout(uint8_t *slot, uint8_t *linear, uint32_t luma_w, uint32_t luma_h, int to_linear)
{
    const uint32_t L = luma_w * luma_h, Cb = L / 4;
    for (uint32_t i = 0; i < L + 2 * Cb; i += 4) {
        uint8_t *t = slot + linear_to_tiled(luma_w / 16, L, Cb, i);
        if (to_linear)
            memcpy(linear + i, t, 4);
        else
            memcpy(t, linear + i, 4);
    }
}

} // extern "C"

// audio_kernel: n_chunks workgroups (time slices) per stream; ring / vpos are updated in place for the
// caller (the kernel writes them to the alternate buffers, emulated with a copy).  Between two
// barriers the kernel's waves run different phases concurrently; the emulation runs them in the
// order that would expose a hazard (the DCTs of step s+1 BEFORE the windows of step s).
template <bool kFma, int kFormat> static void emu_audio_blocks(const AudioArgs &a)
{
    std::vector<float> lds(kAudioLdsFloats);
    for (uint32_t blk = 0; blk < a.n_streams * a.n_chunks; blk++) {
        const uint32_t stream = blk / a.n_chunks, chunk = blk % a.n_chunks;
        const int32_t vpos0 = a.vpos[stream];
        uint32_t tg0, tg1;
        audio_slice_range(a, chunk, vpos0, tg0, tg1);
        if (tg0 >= tg1)
            continue;
        const bool ends_launch = tg1 == a.n_frames * 36;
        if (a.active && a.active[stream] == 0) {
            if (ends_launch)
                for (int tid = 0; tid < kAudioThreads; tid++)
                    audio_carry_state(a, stream, tid);
            continue;
        }
        for (auto &x : lds)
            x = 1e30f; // poison
        const int32_t base0 = audio_step_base0(vpos0, tg0);
        const uint32_t n_steps = audio_step_count(base0, tg1);
        for (int tid = 0; tid < kAudioThreads; tid++) {
            audio_store_window(a, tid, lds.data());
            audio_phase_fetch(a, stream, base0, tg0, tg1, 0, tid, lds.data());
            if (tg0 == 0)
                audio_load_state(a, stream, vpos0, tid, lds.data());
            else
                audio_phase_warmup(a, stream, tg0, tid, lds.data());
        }
        for (int tid = 0; tid < kAudioThreads; tid++)
            audio_phase_dct(a, stream, base0, tg0, tg1, 0, tid, lds.data());
        for (uint32_t si = 0; si < n_steps; si++) {
            for (int tid = 0; tid < kAudioThreads; tid++)
                audio_phase_dct(a, stream, base0, tg0, tg1, si + 1, tid, lds.data());
            for (int tid = 0; tid < kAudioThreads; tid++)
                audio_phase_window<kFma, kFormat>(a, stream, vpos0, base0, tg0, tg1, si, tid, lds.data());
        }
        if (ends_launch) {
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

// the time slices of a launch (audio_slice_range), for the test that checks they tile it
void emu_audio_slice_range(uint32_t n_frames, uint32_t n_chunks, int32_t vpos0, uint32_t chunk, uint32_t *tg0, uint32_t *tg1)
{
    AudioArgs a = {};
    a.n_frames = n_frames;
    a.n_chunks = n_chunks;
    audio_slice_range(a, chunk, vpos0, *tg0, *tg1);
}

// scalar helpers exposed for unit tests
uint32_t emu_avg4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return avg4_u8x4(a, b, c, d); }
uint32_t emu_avg2(uint32_t a, uint32_t b) { return avg_ceil_u8x4(a, b); }
uint32_t emu_xcd_chunk(uint32_t b, uint32_t n) { return xcd_chunk(b, n); }
// The transposition through LDS (rc_tpose_store / rc_tpose_load, the kernel's rc_transpose8_lds): v[64 lanes][8] in place, half a wave
// at a time over ONE buffer of kRcTposeBytes; touched[dword] = 1 + the number of the store instruction (r) that wrote it last.
uint32_t emu_tpose(int32_t *v, uint8_t *touched)
{
    alignas(16) int32_t T[kRcTposeBytes / 4];
    for (int h = 0; h < 2; h++) {
        for (int i = 0; i < kRcTposeBytes / 4; i++)
            T[i] = 0x5a5a5a5a;
        for (int lane = 32 * h; lane < 32 * h + 32; lane++) {
            int32_t x[8];
            memcpy(x, v + lane * 8, sizeof(x));
            rc_tpose_store(T, lane, x);
            for (int r = 0; r < 8; r++) // where instruction r of this lane went: found by its value
                if (touched && h == 0)
                    touched[((lane >> 3) & 3) * kRcTposeStride + (lane & 7) + r * 8] = (uint8_t)(1 + r);
        }
        for (int lane = 32 * h; lane < 32 * h + 32; lane++) {
            int32_t x[8];
            rc_tpose_load(T, lane, x);
            memcpy(v + lane * 8, x, sizeof(x));
        }
    }
    return kRcTposeBytes;
}
uint32_t emu_ycbcr(uint32_t y, uint32_t cb, uint32_t cr) { return ycbcr_to_rgba(y, cb, cr); }

}
