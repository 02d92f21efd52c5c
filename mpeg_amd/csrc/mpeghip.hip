// mpeghip.hip — gfx950 kernels and the C ABI of include/mpeghip.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
//        -I include -I mpeg_amd/csrc mpeg_amd/csrc/mpeghip.hip -o mpeg_amd/libmpeghip.so
//
// There is no CPU path in this library.  Every entry point that needs the GPU
// fails with MPEGHIP_ERR_NO_DEVICE / MPEGHIP_ERR_HIP when it is not there.
#include <hip/hip_runtime.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <new>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "audio_lane.h"
#include "iso11172_synth_window.h"
#include "mpeghip.h"
#include "video_lane.h"
#include "video_pack_lane.h"
#include "video_recon_lane.h"

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

// The 8 x 8 transposition through LDS, in two halves of 4 blocks (video_recon_lane.h: rc_tpose_store / rc_tpose_load): DS operations
// of a wave execute in issue order, so the second half's stores follow the first half's loads without a wait in between.
static __device__ __forceinline__ void rc_transpose8_lds(int32_t *T, int lane, int32_t (&v)[8])
{
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if ((lane >> 5) == h)
            rc_tpose_store(T, lane, v);
        wave_lds_handoff();
        if ((lane >> 5) == h)
            rc_tpose_load(T, lane, v);
        wave_lds_handoff();
    }
}

// ---- reconstruction (video_recon_lane.h has the whole story): one wave = one chunk of 4 macroblocks,
// wave-private LDS, no barrier.  kRgba: the instance for batches with MPEGHIP_PIC_RGBA pictures
// (Frame.RGBA() fused); the other one carries none of that code.  kT16 (video_recon_lane.h, "the wave's coefficient tile"):
// true = the 8 x 8 transposition across lanes by DPP — the instance for typical batches; false = through LDS in two halves, and the
// short dequantisation of dense units under the default matrix — the instance for batches with dense units, which are bound by
// vector-ALU issue (launch_batch picks).  Both on the int16 tile, 8 waves per SIMD.
#ifdef MPG_PHASE_TIMING // instrumented build for tools/phase_timing.py only: s_memtime at the phase boundaries
__device__ uint64_t g_phase_dump[60000 * 8];
#define MPG_STAMP(k) ts[k] = __builtin_readcyclecounter()
#else
#define MPG_STAMP(k)
#endif

// Chunks per wave: ONE.  Rounds 3 - 4 ran two consecutive chunks per wave in the int32-tile instance when a launch had more waves
// than the device has slots (the second header arrived with the first, the hand-over between two workgroups was paid half as
// often: dense +1.4 % then).  With round 5's prologue that no longer pays at any size — 1024 pictures per launch: within 0.2 %,
// with Frame.RGBA fused 1 - 1.5 % SLOWER; 4 - 16 pictures per launch: 8 - 18 % slower, the launch lasts two chunks' chains with
// half the waves (profiles/round5_k_ab_chunks_per_wave_by_launch_size.txt) — and the second form is gone.
#ifndef MPG_CHUNK_AHEAD
#define MPG_CHUNK_AHEAD 256 // chunks; 0 = off (profiles/r3f_ab_chunk_pull_ahead.txt: 64 / 256 / 1024)
#endif
// The arguments a wave needs before its chunk header is back come as scalars in front of the struct: built with
// -mllvm -amdgpu-kernarg-preload-count=14 (mpeg_amd/_build.py) they arrive in SGPRs with the wave instead of through three
// dependent rounds of scalar loads (grid size -> chunk count -> chunk pointer), and the grid size is an argument because the
// hidden one cannot be preloaded.  profiles/r34_ab_kernarg_preload.txt: typical +0.75 %, one picture 10.15 -> 9.86 us.
template <int WAVES, bool kRgba, bool kT16>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void recon_kernel(
    const uint32_t grid8, const uint32_t n_chunks, const uint32_t *const chunks, const uint32_t *const words, const uint8_t *const qmat,
    uint8_t *const frames_b, const uint32_t mb_w, const uint32_t luma_bytes, uint8_t *const rgba, const uint64_t rgba_stride,
    const uint32_t width, const uint32_t height)
{
    static_assert(WAVES == 1, "one wave per workgroup: the wave's LDS starts at 0 (lds_read32x2 / dma_table_and_windows)");
    // Everything a wave needs of the launch: the first 14 dwords arrive preloaded in SGPRs; the plane geometry follows from mb_w and
    // luma_bytes where a rare path wants it (a gathered window, a chunk that is not a run).
#ifdef MPG_PRIO_EARLY
    __builtin_amdgcn_s_setprio(MPG_PRIO_EARLY);
#endif
    VideoArgs a;
    a.frames = nullptr; // (waves address frames through frames_b + the chunk's stream offset only)
    a.frames_b = frames_b;
    a.frame_stride = 0;
    a.mb_w = mb_w;
    a.mb_h = 0;
    a.luma_w = mb_w * 16;
    a.luma_h = 0;
    a.chroma_w = mb_w * 8;
    a.chroma_h = 0;
    a.luma_bytes = luma_bytes;
    a.chroma_bytes = luma_bytes >> 2;
    a.pics = nullptr;
    a.chunks = chunks;
    a.words = words;
    a.qmat = qmat;
    a.n_chunks = n_chunks;
    a.width = width;
    a.height = height;
    a.rgba = rgba;
    a.rgba_stride = rgba_stride;
#ifdef MPG_PHASE_TIMING
    uint64_t ts[8];
#endif
    MPG_STAMP(0);
    constexpr int kLdsBytes = rc_lds_bytes<kT16>();
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[WAVES * kLdsBytes];
    const int lane_all = (int)threadIdx.x;
    // (PERSISTENT waves — as many one-wave workgroups as the device has slots, each taking chunk after chunk of its XCD's range from
    // ticket counters, in order — fill every slot all the time and are 4 - 8 % SLOWER: profiles/round5_d_ab_persistent_waves.txt.)
    uint8_t *lds = lds_all;
    int16_t *T16 = reinterpret_cast<int16_t *>(lds + kRcTileAt); // the int16 tile (and, for !kT16, the transposition buffer over it)
    uint32_t ahead = 0;
    bool ahead_pending = false;
#ifdef MPG_PROBE_PAIRS
    uint64_t probe_read_bias = 0;
#endif
    auto one_chunk = [&](const RcChunk &c, const uint32_t chunk) {
    (void)chunk; // (the instrumented build's stamps)
    // what depends on the lane only: worked out while the header is on its way
    const int lane = lane_all;
    const RcLane k = rc_lane(a, lane);
    const uint32_t n_blocks = rc_n_blocks(c);
#ifdef MPG_PHASE_TIMING
    if (n_blocks > 24) // (never: makes the stamp wait for the chunk)
        return;
#endif
    MPG_STAMP(1);
    const uint32_t *const wbase = rc_word_base(a, c); // the chunk's block words; its entries n_blocks dwords further on
#ifdef MPG_PROBE_PAIRS
    // (probe, tools/ab: waves 2j and 2j + 1 of an XCD's range take chunk j of the launch's first and second half, and the second
    // half READS the first half's frames — what two B pictures of one stream between the same anchors, launched together, would
    // do to the windows' L2 hit rate.  Results are right only while both halves hold the same frames, as bench.py's streams do.)
    uint8_t *const fbase = rc_frame_base(a, c) - probe_read_bias;
#else
    uint8_t *const fbase = rc_frame_base(a, c);       // the stream's frames (biased: kRcDmaBias)
#endif
    uint32_t e = load32_uncounted(wbase, rc_ent_lane_offset(c, 0, lane));
    uint32_t bw = load32_uncounted(wbase, rc_blk_lane_offset(0, lane)); // (pass 0's block words)
    if (lane < kRcWinLanes) {
        const uint32_t off[5] = {rc_table_lane_offset(c, lane), rc_win_offset(c, 0, k), rc_win_offset(c, 1, k), rc_win_offset(c, 2, k),
                                 rc_win_offset(c, 3, k)};
        dma_table_and_windows<kRcQtabAt, kRcWinAt, kRcWinAt + kRcWinBytes, kRcWinAt + 2 * kRcWinBytes, kRcWinAt + 3 * kRcWinBytes>(a.qmat, fbase, off,
                                                                                                                              lds, lane);
    }
    // windows that leave their plane (the reference reads on linearly: rare in streams, 1 - 6 % of the bench's macroblocks): gathered
    // by one-dword loads straight into the window's place, behind the loads above and in flight with them
    if (rc_any_slow(c)) {
        if (lane < kRcGatherLanes) {
            if (c.r[0][0] & kRSlow)
                rc_gather_to_lds<0>(a, c, fbase, lds, lane);
            if (c.r[1][0] & kRSlow)
                rc_gather_to_lds<1>(a, c, fbase, lds, lane);
            if (c.r[2][0] & kRSlow)
                rc_gather_to_lds<2>(a, c, fbase, lds, lane);
            if (c.r[3][0] & kRSlow)
                rc_gather_to_lds<3>(a, c, fbase, lds, lane);
        }
    }
    MPG_STAMP(2);
#ifdef MPG_PRIO_EARLY // (probe: issue priority while a wave sets up and issues its loads; profiles/round6_e_*)
    __builtin_amdgcn_s_setprio(0);
#endif

    int32_t v[8];
    bool table_flat = false;
    uint32_t ent_at = 0, bw_next = 0, e_next = 0;
    i32x4_a4 dense_next = {{0, 0, 0, 0}};
    // step 2: residual pass over coded blocks 8 * pass .. 8 * pass + 7; leaves lane (g, j) with row j of block g
    auto residual_pass = [&](uint32_t pass) {
        const uint32_t np = rc_pass_entries(c, pass);
        if (pass > 0)
            bw = bw_next;
        if (pass > 0)
            e = e_next;
        if ((pass + 1) * 8 < n_blocks) { // the next pass's block words and first 64 entries: on their way while this pass runs
            bw_next = wbase[rc_blk_lane_offset(pass + 1, lane) / 4];
            if (rc_pass_entries(c, pass + 1)) // (a pass of dense units has none)
                e_next = *rc_ent_src(a, c, ent_at + np, lane); // (beyond that pass's entries: ignored; the array is padded)
        }
        const bool mine = pass * 8 + ((uint32_t)lane >> 3) < n_blocks;
        auto scatter_entries = [&](auto &&scatter_one) {
            for (uint32_t r = 0; r < np; r += 64) {
                if (r > 0)
                    e = *rc_ent_src(a, c, ent_at + r, lane);
                if (r + (uint32_t)lane < np)
                    scatter_one(e);
            }
            ent_at += np;
        };
        auto zero_columns = [&]() {
#pragma unroll
            for (int r = 0; r < 8; r++)
                v[r] = 0; // (lanes beyond the last block: their result is not used)
        };
        // blocks that travel as dense units: their columns come straight from the unit, dequantised in place of the tile
        // read.  (The next pass's columns: its block words are here by now, the units' 16 bytes per lane arrive while this
        // pass finishes — otherwise every pass waits out a dependent HBM read.)
        auto dense_columns = [&]() {
            if (!rc_any_dense(c))
                return;
            const bool dense_here = mine && (bw & kBDense);
            // the default non-intra matrix and no intra unit among the pass's: the short dequantisation (wave-uniform choice).
            // In the !kT16 instance only, the one that batches of dense units run on: typical batches have a dense
            // unit here and there, and the int16-tile instance lost 0.5 % to carrying the second form
            // (profiles/r25_ab_dense_default_non_intra_matrix.txt).
            const bool flat = !kT16 && table_flat && all_in_wave(!dense_here || (int32_t)bw < 0);
            if (dense_here) {
                const i32x4_a4 lv = pass > 0 ? dense_next : rc_dense_read(a, c, bw, lane);
                if (flat)
                    rc_dense_cols<true>(lv, lds, bw, lane, v);
                else
                    rc_dense_cols<false>(lv, lds, bw, lane, v);
            }
            if ((pass + 1) * 8 + ((uint32_t)lane >> 3) < n_blocks && (bw_next & kBDense))
                dense_next = rc_dense_read(a, c, bw_next, lane);
        };
        // the int16 tile holds sparse blocks only: a pass without entries does not go through it
        if (np) {
            rc_zero_tile16(T16, lane);
            wave_lds_handoff();
            scatter_entries([&](uint32_t ent) { rc_scatter16(T16, lds, ent); });
            wave_lds_handoff();
            rc_cols_load16(T16, lds, lane, v);
        } else {
            zero_columns();
        }
        if (rc_any_special(c)) { // (one test for the common chunk — predicted macroblocks, sparse blocks — instead of three)
            if (rc_any_dcword(c) && mine)
                rc_dc_from_word(bw, lane, v);
            if (rc_any_raw(c) && mine && (bw & kBRaw)) // int32 snapshot blocks (damaged streams): as they are, from HBM
                rc_raw_cols(a, c, bw, lane, v);
            dense_columns();
        }
        idct8<false>(v);
        // column j -> row j: across the block's 8 lanes by DPP, or — the instance for batches of dense units, which are bound by
        // vector-ALU issue — through the tile, which every lane has read by now
        if (kT16)
            rc_transpose8(v, lane);
        else
            rc_transpose8_lds(reinterpret_cast<int32_t *>(T16), lane, v);
        idct8<true>(v);
    };
    // step 4: residual rows onto the prediction
    auto add_residual = [&](uint32_t pass) {
        wave_lds_handoff();
        if (pass * 8 + ((uint32_t)lane >> 3) < n_blocks)
            rc_rmw(lds, bw, lane, v);
    };
    if (n_blocks) {
        wait_loads<4>(); // the entries, the block words and the table are there; the four windows may still be on their way
        settle(e);
        settle(bw);
        wave_lds_handoff();
        if (!kT16 && rc_any_dense(c)) // is the stream's non-intra matrix the default one?  (lane (g, j) looks at column j)
            table_flat = all_in_wave(rc_non_intra_column_flat(lds, lane));
        residual_pass(0);
    }
    MPG_STAMP(3);
    // step 3: motion compensation, half-pel modes wave-uniform per macroblock: window in LDS -> O_m over it
    wait_loads<0>(); // all four windows are there
#ifdef MPG_PRIO_LATE // (probe: issue priority once a wave has all its data — it only computes and stores from here)
    __builtin_amdgcn_s_setprio(MPG_PRIO_LATE);
#endif
    settle(e);       // (on every path: until here the entries' and block words' registers belong to loads in flight)
    settle(bw);
#if MPG_CHUNK_AHEAD
    if (ahead_pending)
        settle(ahead);
    ahead_pending = false;
#endif
    wave_lds_handoff();
    // (every macroblock of the common kind — predicted, window inside its plane — takes the short way: its record's scalars are
    // what the instructions consume; intra macroblocks put zeros, a window that leaves its plane is gathered first)
    auto mc_one = [&](auto M) {
        constexpr int m = decltype(M)::value;
        const uint32_t r0 = c.r[m][0];
        uint32_t yl = 0, yc = 0;
        if (!(r0 & (kRIntra | kRDead | kRSlow))) {
            yl = rc_mc_luma<m>(lds, k, r0, c.r[m][3]);
            yc = rc_mc_chroma<m>(lds, k, lane, r0, c.r[m][4], c.r[m][5]);
        } else {
            if (r0 & kRDead)
                return;
            if (r0 & kRSlow) { // the window leaves its plane: the reference's linear reads, gathered in step 1 (rare)
                const uint8_t *win = lds + rc_win_at(m);
                yl = rc_mc_luma_slow(win, lane, r0, c.r[m][3], k.ones);
                yc = rc_mc_chroma_slow(win, lane, r0, c.r[m][4], k.ones);
            }
        }
        wave_lds_handoff(); // every lane has its taps
        uint8_t *win = lds + rc_win_at(m);
        *reinterpret_cast<uint32_t *>(win + k.out_luma) = yl;
        *reinterpret_cast<uint32_t *>(win + k.out_chroma) = yc; // (lanes 32..63 repeat lanes 0..31: the same bytes to the same place)
    };
    mc_one(std::integral_constant<int, 0>{});
    mc_one(std::integral_constant<int, 1>{});
    mc_one(std::integral_constant<int, 2>{});
    mc_one(std::integral_constant<int, 3>{});
    MPG_STAMP(4);
    if (n_blocks) {
        add_residual(0);
        for (uint32_t pass = 1; pass * 8 < n_blocks; pass++) {
            residual_pass(pass);
            add_residual(pass);
        }
    }
    MPG_STAMP(5);
    wave_lds_handoff();
    // step 5
    const bool run = (c.h[4] & kCRun) != 0;
    const bool rgba = kRgba && (c.h[4] & kCRgba) != 0;
    const uint32_t n_live = rc_n_live(c);
    if (run) {
        rc_store_run(a, c, lane, lds);
    } else {
#pragma unroll
        for (uint32_t m = 0; m < (uint32_t)kRcMbs; m++)
            if (m < n_live)
                rc_store_mb(a, c, m, lane, lds, rgba);
    }
    if (kRgba && rgba) {
        if (!run)
            wave_lds_handoff();
        uint8_t *const img = rc_rgba_image(a, c);
        if (run && rc_run_in_one_row(c)) {
#pragma unroll
            for (uint32_t q = 0; q < 4; q++)
                rc_rgba_run_rows(a, c, img, q, lane, lds);
        } else { // (also a run that wraps a row end: its planes left as a run, its image rows are per macroblock)
#pragma unroll
            for (uint32_t m = 0; m < (uint32_t)kRcMbs; m++)
                if (m < n_live)
                    rc_rgba_mb(a, c, img, m, lane, lds);
        }
    }
#ifdef MPG_PHASE_TIMING
    MPG_STAMP(6);
    if (lane == 0 && chunk < 60000) {
        for (int i = 0; i < 7; i++)
            g_phase_dump[(uint64_t)chunk * 8 + i] = ts[i];
        g_phase_dump[(uint64_t)chunk * 8 + 7] = n_blocks;
    }
#endif
    };
    // pull the chunk a later wave of this XCD's range will take towards L2 (a chunk is one 128-byte line) so that that wave's
    // scalar loads find them there; nothing is done with the data.  (Also pulling those chunks' first words, by a dependent
    // load once the header is here, gains nothing: profiles/r3g_ab_pull_ahead_distance_and_words.txt.)
    auto pull_ahead = [&](uint32_t first, uint32_t end) {
#if MPG_CHUNK_AHEAD
        const uint32_t step = first + MPG_CHUNK_AHEAD + 1 <= end ? MPG_CHUNK_AHEAD * kRcChunkDwords * 4 : 0u;
        ahead = load32_uncounted(reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(a.chunks) + first * (kRcChunkDwords * 4)), step);
        ahead_pending = true;
#else
        (void)first;
        (void)end;
#endif
    };
    // XCD-aware remap (block b runs on XCD b % 8; each XCD has its own L2): every XCD gets one contiguous range of chunks.  The
    // grid is a multiple of 8 (launch_batch rounds it up: at most 7 waves find nothing to do), so the map is a multiply-add.
#ifdef MPG_PROBE_PAIRS
    const uint32_t logical = __builtin_amdgcn_readfirstlane(xcd_chunk(blockIdx.x, grid8));
    if (logical >= a.n_chunks)
        return;
    const uint32_t first = (logical & 1u) * (a.n_chunks >> 1) + (logical >> 1);
#ifndef MPG_PROBE_PAIRS_NOSHARE // (the control: the same order of chunks, every stream reads its own frames)
    probe_read_bias = (logical & 1u) ? rgba_stride : 0; // (launch_batch passes half the streams' frames in this argument)
#endif
#else
    const uint32_t first = __builtin_amdgcn_readfirstlane(xcd_chunk(blockIdx.x, grid8));
    if (first >= a.n_chunks)
        return;
#endif
    pull_ahead(first, a.n_chunks);
    // step 1: one round of scalar loads (the chunk), then its vector loads
    const RcChunk c0 = rc_load_chunk(a, first);
    one_chunk(c0, first);
}
#undef MPG_STAMP

// ---- the same work for launches that leave most wave slots empty — ONE picture (BASELINE config 3 as written: 2 040 chunks for
// 8 192 slots): FOUR waves per chunk.  Such a launch lasts one chunk's chain of dependent steps, and a single wave walks through
// ~700 instructions and ~40 LDS round trips with one other wave per SIMD to hide them (profiles/round5_e_*: 9.4 us per picture over
// a launch floor of 3.0).  Here wave w fetches window w and runs residual pass w (a chunk has at most three) and the motion
// compensation of macroblock w, all at the same time; two workgroup barriers order the hand-over of the output bytes (residual
// rows are added by the wave that made them; then every wave stores its share: luma / chroma of a run, four image rows each of
// a fused colour conversion).  The lane functions, the device format and the arithmetic are recon_kernel's own (int16 tile);
// launch_batch takes this kernel when four waves per chunk still fit the device's slots and no instance is pinned.
// kMirror (mpeghip_video_host_mirror: a lone decoder's store): every macroblock written is also written, linearly, into the frame's
// copy in pinned host memory — the frame Video.Decode returns is then in the caller's hands when the launch is over, with no
// untiling launch behind it (rc_mirror_mb).
template <bool kRgba, bool kMirror>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void recon_wide_kernel(
    const uint32_t grid8, const uint32_t n_chunks, const uint32_t *const chunks, const uint32_t *const words, const uint8_t *const qmat,
    uint8_t *const frames_b, const uint32_t mb_w, const uint32_t luma_bytes, uint8_t *const rgba, const uint64_t rgba_stride,
    const uint32_t width, const uint32_t height)
{
    static_assert(!(kRgba && kMirror), "the mirrored instance carries no colour conversion (launch_batch)");
    // (the mirroring instance converts no colours: the two arguments that name the RGBA images name the host mirror there, and the
    // kernel's signature — its preloaded arguments — is the same for every instance)
    uint8_t *const mirror = rgba;
    const uint64_t mirror_stride = rgba_stride;
    VideoArgs a;
    a.frames = nullptr;
    a.frames_b = frames_b;
    a.frame_stride = 0;
    a.mb_w = mb_w;
    a.mb_h = 0;
    a.luma_w = mb_w * 16;
    a.luma_h = 0;
    a.chroma_w = mb_w * 8;
    a.chroma_h = 0;
    a.luma_bytes = luma_bytes;
    a.chroma_bytes = luma_bytes >> 2;
    a.pics = nullptr;
    a.chunks = chunks;
    a.words = words;
    a.qmat = qmat;
    a.n_chunks = n_chunks;
    a.width = width;
    a.height = height;
    a.rgba = rgba;
    a.rgba_stride = rgba_stride;
    // table | four windows (-> O_m) as in recon_kernel, then one int16 tile per pass
    __shared__ __attribute__((aligned(16))) uint8_t lds[kRcTileAt + 3 * kRcTileBytes16];
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t chunk = __builtin_amdgcn_readfirstlane(xcd_chunk(blockIdx.x, grid8));
    if (chunk >= a.n_chunks)
        return; // (the whole workgroup: no barrier is left waiting)
    const RcChunk c = rc_load_chunk(a, chunk);
    const RcLane k = rc_lane(a, lane);
    const uint32_t n_blocks = rc_n_blocks(c);
    const bool my_pass = w * 8 < n_blocks;
    const uint32_t *const wbase = rc_word_base(a, c);
    uint8_t *const fbase = rc_frame_base(a, c);
    // what a record names by its macroblock is reached through a switch on the (wave-uniform) wave number: an index into the
    // chunk's scalars would move the whole chunk to scratch
    auto by_wave = [&](auto &&f) {
        switch (w) {
        case 0: f(std::integral_constant<int, 0>{}); break;
        case 1: f(std::integral_constant<int, 1>{}); break;
        case 2: f(std::integral_constant<int, 2>{}); break;
        default: f(std::integral_constant<int, 3>{}); break;
        }
    };
    // step 1: the wave's loads — its pass's first 64 entries and block words, the table (every wave asks for the same 192 bytes:
    // no barrier in front of the passes), its window
    uint32_t ent_at = 0, np = 0;
    if (w == 0)
        np = rc_pass_entries(c, 0);
    if (w == 1)
        ent_at = rc_pass_entries(c, 0), np = rc_pass_entries(c, 1);
    if (w == 2)
        ent_at = rc_pass_entries(c, 0) + rc_pass_entries(c, 1), np = rc_pass_entries(c, 2);
    uint32_t e = 0, bw = 0;
    if (my_pass) {
        e = load32_uncounted(wbase, rc_ent_lane_offset(c, ent_at, lane));
        bw = load32_uncounted(wbase, rc_blk_lane_offset(w, lane));
    }
    if (lane < kRcQtabBytes / kRcPiece) // (its own lanes only: another wave's window must not be written over)
        dma16_to_lds<kRcQtabAt>(a.qmat, rc_table_lane_offset(c, lane), lds, lane);
    if (lane < kRcWinLanes)
        by_wave([&](auto M) {
            constexpr int m = decltype(M)::value;
            dma16_to_lds<kRcWinAt + m * kRcWinBytes>(fbase, rc_win_offset(c, m, k), lds, lane);
            if ((c.r[m][0] & kRSlow) && lane < kRcGatherLanes) // (recon_kernel: a window that leaves its plane, gathered behind it)
                rc_gather_to_lds<m>(a, c, fbase, lds, lane);
        });
    // step 2: residual pass w (recon_kernel's int16-tile form) -> row j of block g in lane (g, j)
    int32_t v[8];
    const bool mine = w * 8 + ((uint32_t)lane >> 3) < n_blocks;
    if (my_pass) {
        int16_t *T16 = reinterpret_cast<int16_t *>(lds + kRcTileAt + w * kRcTileBytes16);
        wait_loads<1>(); // the entries, the block words and the table are there; the window may still be on its way
        settle(e);
        settle(bw);
        wave_lds_handoff();
        if (np) {
            rc_zero_tile16(T16, lane);
            wave_lds_handoff();
            for (uint32_t r = 0; r < np; r += 64) {
                if (r > 0)
                    e = *rc_ent_src(a, c, ent_at + r, lane);
                if (r + (uint32_t)lane < np)
                    rc_scatter16(T16, lds, e);
            }
            wave_lds_handoff();
            rc_cols_load16(T16, lds, lane, v);
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++)
                v[r] = 0;
        }
        if (rc_any_special(c)) {
            if (rc_any_dcword(c) && mine)
                rc_dc_from_word(bw, lane, v);
            if (rc_any_raw(c) && mine && (bw & kBRaw))
                rc_raw_cols(a, c, bw, lane, v);
            if (rc_any_dense(c) && mine && (bw & kBDense))
                rc_dense_cols<false>(rc_dense_read(a, c, bw, lane), lds, bw, lane, v);
        }
        idct8<false>(v);
        rc_transpose8(v, lane);
        idct8<true>(v);
    }
    // step 3: motion compensation of macroblock w
    wait_loads<0>();
    settle(e);
    settle(bw);
    wave_lds_handoff();
    by_wave([&](auto M) {
        constexpr int m = decltype(M)::value;
        const uint32_t r0 = c.r[m][0];
        if (r0 & kRDead)
            return;
        uint32_t yl = 0, yc = 0;
        uint8_t *win = lds + rc_win_at(m);
        if (!(r0 & (kRIntra | kRSlow))) {
            yl = rc_mc_luma<m>(lds, k, r0, c.r[m][3]);
            yc = rc_mc_chroma<m>(lds, k, lane, r0, c.r[m][4], c.r[m][5]);
        } else if (r0 & kRSlow) {
            yl = rc_mc_luma_slow(win, lane, r0, c.r[m][3], k.ones);
            yc = rc_mc_chroma_slow(win, lane, r0, c.r[m][4], k.ones);
        }
        wave_lds_handoff(); // every lane has its taps
        *reinterpret_cast<uint32_t *>(win + k.out_luma) = yl;
        *reinterpret_cast<uint32_t *>(win + k.out_chroma) = yc;
    });
    workgroup_barrier_lds(); // the four O_m are complete
    // step 4: the wave's residual rows onto them
    if (my_pass && mine)
        rc_rmw(lds, bw, lane, v);
    workgroup_barrier_lds();
    // step 5: stores, every wave its share
    const bool run = (c.h[4] & kCRun) != 0;
    const bool to_rgba = kRgba && (c.h[4] & kCRgba) != 0;
    const uint32_t n_live = rc_n_live(c);
    uint8_t *const mirror_frame = kMirror ? mirror + ((uint64_t)rc_stream(c) * MPEGHIP_SLOTS + rc_cur_slot(c)) * mirror_stride : nullptr;
    if (run) {
        if (w == 0)
            rc_store_run_luma(a, c, lane, lds);
        if (w == 1)
            rc_store_run_chroma(a, c, lane, lds);
        if (kRgba && to_rgba) {
            if (rc_run_in_one_row(c))
                rc_rgba_run_rows(a, c, rc_rgba_image(a, c), w, lane, lds);
            else // a run that wraps a row end: wave w converts macroblock w
                by_wave([&](auto M) { rc_rgba_mb(a, c, rc_rgba_image(a, c), (uint32_t)decltype(M)::value, lane, lds); });
        }
        if (kMirror) // (a run has four live macroblocks, every byte of them written)
            by_wave([&](auto M) { rc_mirror_mb(a, c, mirror_frame, (uint32_t)decltype(M)::value, lane, lds); });
    } else if (w < n_live) {
        by_wave([&](auto M) {
            constexpr uint32_t m = (uint32_t)decltype(M)::value;
            rc_store_mb(a, c, m, lane, lds, to_rgba || kMirror);
            if (kRgba && to_rgba) {
                wave_lds_handoff();
                rc_rgba_mb(a, c, rc_rgba_image(a, c), m, lane, lds);
            }
            if (kMirror) {
                wave_lds_handoff(); // (the pixels an invalid intra block keeps are in O_m now)
                rc_mirror_mb(a, c, mirror_frame, m, lane, lds);
            }
        });
    }
}

// Which 4x2 pixel block a thread of the whole-frame conversions takes.  The frame store is tiled (video_lane.h), so a
// wave walks TILES, not picture rows: wave = 4 macroblocks side by side x 8 rows, lane = (macroblock lane>>4, row pair
// (lane>>2)&3, quad lane&3): its luma reads are 4 x 128 contiguous bytes, each RGBA store instruction writes one row of
// 64 pixels = 256 contiguous bytes.  Workgroup = 4 waves = 8 macroblocks x 16 rows; grid (mb_w / 8, mb_h, frames).
MPG_HD void rgba_thread_quad(uint32_t tid, uint32_t bx, uint32_t by, uint32_t &x4, uint32_t &yp)
{
    const uint32_t wave = tid >> 6, lane = tid & 63;
    const uint32_t mb = (bx * 2 + (wave >> 1)) * 4 + (lane >> 4);
    x4 = mb * 4 + (lane & 3);
    yp = by * 8 + (wave & 1) * 4 + ((lane >> 2) & 3);
}

// Frame.RGBA of the cur slot of every picture flagged MPEGHIP_PIC_RGBA: grid (macroblock columns / 8, macroblock rows, pictures).
__global__ __launch_bounds__(256) void rgba_pics_kernel(const VideoArgs a, uint32_t pic0)
{
    const mpeghip_pic_desc p = a.pics[pic0 + blockIdx.z];
    if (!(p.flags & MPEGHIP_PIC_RGBA))
        return;
    uint32_t x4, y;
    rgba_thread_quad(threadIdx.x, blockIdx.x, blockIdx.y, x4, y);
    const uint64_t fs = (uint64_t)p.stream * MPEGHIP_SLOTS + p.cur;
    rgba_convert_quad(a.frames + fs * a.frame_stride, a.mb_w, a.luma_bytes, a.chroma_bytes, a.width, a.height, x4, y,
                      a.rgba + fs * a.rgba_stride);
}

// Frame.RGBA for whole slots: grid (macroblock columns / 8, macroblock rows, streams).
__global__ __launch_bounds__(256) void rgba_kernel(const uint8_t *frames, uint64_t frame_stride,
                                                  uint8_t *rgba, uint64_t rgba_stride,
                                                  uint32_t mb_w, uint32_t luma_bytes, uint32_t chroma_bytes,
                                                  uint32_t width, uint32_t height,
                                                  uint32_t slot, uint32_t stream0)
{
    uint32_t x4, y;
    rgba_thread_quad(threadIdx.x, blockIdx.x, blockIdx.y, x4, y);
    const uint64_t fs = (uint64_t)(stream0 + blockIdx.z) * MPEGHIP_SLOTS + slot;
    rgba_convert_quad(frames + fs * frame_stride, mb_w, luma_bytes, chroma_bytes, width, height, x4, y, rgba + fs * rgba_stride);
}

// Replicate a one-stream batch for streams 1..n-1 (benchmark batches): stream s gets its own copy of the
// chunks, shifted to its frames, its table and its own copy of the words.  Upload-time scaffolding of
// mpeghip_video_batch_upload_replicated, never inside a timed region.
// ---- the device-side packer (video_pack_lane.h has the whole story): one wave = 64 consecutive macroblocks = 16 chunks of one
// picture of a device-packed stage; grid = pictures x (waves the largest picture needs).
__global__ __launch_bounds__(64) void pack_kernel(const PackArgs a)
{
    // LDS: the exchange array, then the window — a.win_dwords of it, sized by the host for the commit's pictures (words per
    // macroblock): 4 096 dwords for a commit with an I picture in it (2 waves per SIMD), a quarter of that for typical ones
    extern __shared__ __attribute__((aligned(16))) uint32_t pack_lds[];
    uint32_t *const xch = pack_lds;
    uint32_t *const win = pack_lds + 64 * kPkXchDwords;
    uint32_t *const wout = win + a.win_dwords; // the window of the words the wave produces
    const uint32_t pic = blockIdx.x / a.groups_per_pic, g = blockIdx.x - pic * a.groups_per_pic;
    const mpeghip_pic_desc p = a.pics[pic];
    if (g * 64 >= p.mb_count)
        return;
    const PkPic x = a.aux[pic];
    const int lane = (int)threadIdx.x;
    const uint32_t k = g * 64 + (uint32_t)lane;
    // the lane's descriptor, and where the next wave's data begins (= where this wave's ends, by the order rule)
    const uint32_t k_next = g * 64 + 64;
    mpeghip_mb_desc mb;
    memset(&mb, 0, sizeof(mb));
    if (k < p.mb_count)
        mb = a.mbs[p.mb_first + k];
    const uint32_t next_coef_off = k_next < p.mb_count ? a.mbs[p.mb_first + k_next].coef_off : 0u;
    // phase 0: the wave's window of the picture's words -> LDS, 16 bytes per lane and load, all loads in flight at once
    PkWin in;
    PkOut out;
    in.glob = a.words_in + x.word_first;
    in.lds = win;
    out.glob = a.words_out;
    out.lds = wout;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb.coef_off); // (lane 0 has a macroblock)
    {
        uint32_t lo4, n;
        pk_window_range(lo, k_next < p.mb_count ? next_coef_off : x.n_words, x.n_words, a.win_dwords, lo4, n);
        in.lo = lo4;
        in.n = n;
        out.base = x.word_first + lo4;
        out.n = n;
        for (uint32_t i = (uint32_t)lane * 4; i < n; i += 256) { // (a picture's words begin on a 64-byte boundary; the buffer is padded)
            *reinterpret_cast<u32x4 *>(win + i) = *reinterpret_cast<const u32x4 *>(in.glob + lo4 + i);
            // (the produced words' window starts out zeroed: the gaps between chunks' words leave it as they are — no
            // uninitialised LDS reaches device memory)
            *reinterpret_cast<u32x4 *>(wout + i) = u32x4{{0, 0, 0, 0}};
        }
    }
    wave_lds_handoff();
    const PkLane L = pk_scan(a, pic, p, x, k, mb, in);
    // which of its two references the picture's macroblocks read (the dependency check between pictures, pack_gate_kernel)
    const uint32_t use = L.ok ? L.use : 0u;
    const bool fwd = __ballot((use & 1u) != 0) != 0, bwd = __ballot((use & 2u) != 0) != 0;
    if (lane == 0 && (fwd || bwd))
        atomicOr(&a.aux[pic].use, (fwd ? 1u : 0u) | (bwd ? 2u : 0u));
    pk_share(xch, lane, L);
    wave_lds_handoff();
    pk_emit(a, p, x, k, lane, L, xch, next_coef_off, in, out);
    // the produced words leave the window: whole-wave stores, the wave's own territory only (from its first macroblock's
    // offset; the dwords in front of it inside the 16-byte aligned window are the previous wave's)
    wave_lds_handoff();
    for (uint32_t i = (uint32_t)lane + (lo - in.lo); i < out.n; i += 64)
        out.glob[out.base + i] = wout[i];
}

// Behind pack_kernel, one workgroup per picture: is the picture refused — by its own report, or because a macroblock of it reads a
// slot that another picture of its stream in this commit writes (the host lists the candidates: picture, which of its references;
// whether a macroblock reads it, only pack_kernel knows)?  Then every chunk of it becomes a dead one: recon_kernel, next on the
// stream, writes nothing of it.  The last workgroup to finish publishes the verdict where the host finds it (pinned memory):
// [0] the first report in submit order (kPkNoError: none), [1] the number of refused pictures, then their indices.
struct PkDep { uint32_t pic, mask; };
constexpr uint32_t kPkVerdictList = 1020; // refused pictures named per commit (the count is exact beyond that)
__global__ __launch_bounds__(256) void pack_gate_kernel(const unsigned long long *err, const PkPic *aux, const mpeghip_pic_desc *pics,
                                                        const PkDep *deps, uint32_t n_deps, uint32_t *chunks, uint32_t n_pics,
                                                        unsigned long long *scratch /* [0] first report, [1] refused, [2] workgroups done */,
                                                        unsigned long long *verdict)
{
    __shared__ unsigned long long key;
    const uint32_t pic = blockIdx.x;
    if (threadIdx.x == 0)
        key = err[pic];
    __syncthreads();
    unsigned long long mine = kPkNoError;
    for (uint32_t i = threadIdx.x; i < n_deps; i += 256)
        if (deps[i].pic == pic && (aux[pic].use & deps[i].mask)) {
            const unsigned long long k = ((unsigned long long)pics[pic].mb_first << 8) | kPkDepends;
            mine = k < mine ? k : mine;
        }
    if (mine != kPkNoError)
        atomicMin(&key, mine);
    __syncthreads();
    const unsigned long long found = key;
    if (found != kPkNoError) {
        const uint32_t c0 = aux[pic].chunk_first, nc = (pics[pic].mb_count + kRcMbs - 1) / kRcMbs;
        for (uint32_t c = threadIdx.x; c < nc; c += 256)
            rc_make_dead_chunk(chunks + (size_t)(c0 + c) * kRcChunkDwords);
    }
    if (threadIdx.x != 0)
        return;
    if (found != kPkNoError) {
        atomicMin(&scratch[0], found);
        const unsigned long long at = atomicAdd(&scratch[1], 1ull);
        if (at < kPkVerdictList)
            reinterpret_cast<uint32_t *>(verdict + 2)[at] = pic; // (pinned host memory)
    }
    __threadfence_system();
    if (atomicAdd(&scratch[2], 1ull) + 1 == n_pics) { // the last one: everybody's reports are in
        __threadfence();
        verdict[1] = atomicAdd(&scratch[1], 0ull);
        verdict[0] = atomicMin(&scratch[0], kPkNoError);
        __threadfence_system();
    }
}

struct ReplicateSteps {
    uint32_t words;       // words of one stream
    uint64_t frames;      // MPEGHIP_SLOTS * frame_stride: bytes from one stream's first slot to the next one's
};
__global__ void replicate_kernel(mpeghip_pic_desc *pics, uint32_t n_pics, uint32_t *chunks, uint32_t n_chunks,
                                 uint32_t mbs_per_stream, ReplicateSteps k, uint32_t n_streams)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total_chunks = (uint64_t)n_chunks * n_streams;
    if (gid >= (uint64_t)n_chunks && gid < total_chunks) {
        const uint32_t s = (uint32_t)(gid / n_chunks), i = (uint32_t)(gid % n_chunks);
        const u32x4 *src = reinterpret_cast<const u32x4 *>(chunks + (uint64_t)i * kRcChunkDwords);
        u32x4 *dst = reinterpret_cast<u32x4 *>(chunks + gid * kRcChunkDwords);
        u32x4 h0 = src[0], h1 = src[1];
        // (records name their windows from the stream's base: only the header knows which stream this is)
        const uint64_t frames = ((uint64_t)h0.v[0] | ((uint64_t)h0.v[1] << 32)) + (uint64_t)s * k.frames;
        const uint64_t words = ((uint64_t)h0.v[2] | ((uint64_t)h0.v[3] << 32)) + (uint64_t)s * k.words * 4;
        h0.v[0] = (uint32_t)frames, h0.v[1] = (uint32_t)(frames >> 32);
        h0.v[2] = (uint32_t)words, h0.v[3] = (uint32_t)(words >> 32);
        h1.v[1] += s << kHStreamShift;
        dst[0] = h0;
        dst[1] = h1;
        for (int q = 2; q < kRcChunkDwords / 4; q++)
            dst[q] = src[q];
        static_assert(kRcChunkDwords == 32 && kRcHeadDwords == 8, "eight 16-byte quarters per chunk, the header in the first two");
    }
    const uint64_t total_pics = (uint64_t)n_pics * n_streams;
    if (gid >= (uint64_t)n_pics && gid < total_pics) {
        const uint32_t s = (uint32_t)(gid / n_pics), i = (uint32_t)(gid % n_pics);
        mpeghip_pic_desc p = pics[i];
        p.stream = s;
        p.mb_first += s * mbs_per_stream;
        pics[gid] = p;
    }
}

// The reference's linear view of one slot's planes <-> the tiled frame (video_lane.h): one dword per thread.
// to_linear: buf[i] = slot[linear_to_tiled(i)]; else slot[linear_to_tiled(i)] = buf[i], for the byte range [first, first + n).
__global__ __launch_bounds__(256) void relayout_kernel(uint8_t *slot, uint8_t *buf, uint32_t first, uint32_t n_bytes, uint32_t mb_w,
                                                       uint32_t luma_bytes, uint32_t chroma_bytes, int to_linear)
{
    const uint32_t i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n_bytes)
        return;
    uint32_t *t = reinterpret_cast<uint32_t *>(slot + linear_to_tiled(mb_w, luma_bytes, chroma_bytes, first + i));
    uint32_t *l = reinterpret_cast<uint32_t *>(buf + i);
    if (to_linear)
        *l = *t;
    else
        *t = *l;
}

// FNV-1a-64 over Y||Cb||Cr of one slot per stream (mpeg_test.go:221-223); one
// thread per stream — a test aid, not a hot path.
__global__ void hash_kernel(const uint8_t *frames, uint64_t frame_stride, uint32_t slot, uint32_t mb_w, uint32_t luma_bytes,
                            uint32_t chroma_bytes, uint32_t n_streams, uint64_t *out)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams)
        return;
    const uint8_t *p = frames + ((uint64_t)s * MPEGHIP_SLOTS + slot) * frame_stride;
    const uint32_t n_bytes = luma_bytes + 2 * chroma_bytes; // in the reference's order; 8 linear bytes stay together in a tile row
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t i = 0; i < n_bytes; i += 8) {
        uint64_t w = *reinterpret_cast<const uint64_t *>(p + linear_to_tiled(mb_w, luma_bytes, chroma_bytes, i));
        for (int k = 0; k < 8; k++) {
            h ^= (w >> (8 * k)) & 0xff;
            h *= 0x100000001b3ull;
        }
    }
    out[s] = h;
}

template <bool kFma, int kFormat>
__global__ __launch_bounds__(kAudioThreads) void audio_kernel(const AudioArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[kAudioLdsFloats];
    const uint32_t stream = blockIdx.x / a.n_chunks, chunk = blockIdx.x % a.n_chunks;
    const int tid = threadIdx.x;
    const int32_t vpos0 = a.vpos[stream];
    uint32_t tg0, tg1;
    audio_slice_range(a, chunk, vpos0, tg0, tg1);
    if (tg0 >= tg1)
        return; // empty slice (wave-uniform, before any barrier)
    const bool ends_launch = tg1 == a.n_frames * 36;
    if (a.active && a.active[stream] == 0) { // (workgroup-uniform)
        if (ends_launch)
            audio_carry_state(a, stream, tid);
        return;
    }
    const int32_t base0 = audio_step_base0(vpos0, tg0); // steps aligned to the window's position cycle (audio_lane.h)
    const uint32_t n_steps = audio_step_count(base0, tg1);
    audio_store_window(a, tid, lds);
    // Barriers order LDS only; the wave that has direct-to-LDS loads in flight waits for them itself, and nobody waits
    // for output stores (they are never read here).
    auto step_barrier = [&](uint32_t loading_wave) {
        if ((uint32_t)(tid >> 6) == loading_wave)
            wait_loads<0>();
        workgroup_barrier_lds();
    };
    // prologue: samples of step 0 in flight, history from the state or rebuilt
    audio_phase_fetch(a, stream, base0, tg0, tg1, 0, tid, lds);
    if (tg0 == 0)
        audio_load_state(a, stream, vpos0, tid, lds);
    else
        audio_phase_warmup(a, stream, tg0, tid, lds);
    step_barrier(dct_wave(0));
    audio_phase_dct(a, stream, base0, tg0, tg1, 0, tid, lds); // (also puts the samples of step 1 in flight)
    step_barrier(dct_wave(0));
    for (uint32_t si = 0; si < n_steps; si++) {
        // the wave that runs DCT(si + 1) does its one window pair first: its two output stores are then long gone when
        // it waits for its refill loads in front of the barrier
        audio_phase_window<kFma, kFormat>(a, stream, vpos0, base0, tg0, tg1, si, tid, lds);
        audio_phase_dct(a, stream, base0, tg0, tg1, si + 1, tid, lds); // wave (si+1)%4; refills the staging buffer for step si+2
        step_barrier(dct_wave(si + 1));
    }
    if (ends_launch) { // the slice that ends the launch owns the state hand-over
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

// Nothing throws across the C ABI: entry points that allocate on the host (std::vector, new) run their bodies through this.
template <class F> static int no_throw(F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(MPEGHIP_ERR_OOM, "host allocation failed");
    } catch (const std::exception &e) {
        return fail(MPEGHIP_ERR_HIP, "unexpected exception: %s", e.what());
    }
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
    // staged commits send their staging buffer on a stream of their own: the copy of commit N + 1 then runs while the kernels
    // of commit N do (one stream would run copy, kernels, copy, kernels one after the other — the link idle a quarter of the time)
    hipStream_t copy_stream = nullptr;
};

// A batch on the device is ONE allocation: pictures | chunks | words (video_recon_lane.h), each region
// 64-byte aligned; resident batches hold `replicas` copies of every region.
struct mpeghip_batch {
    mpeghip_video *owner = nullptr;
    uint8_t *d_blob = nullptr;
    size_t cap_blob = 0;
    mpeghip_pic_desc *d_pics = nullptr;
    uint32_t *d_chunks = nullptr, *d_words = nullptr;
    uint64_t n_pics = 0, n_mbs = 0, n_chunks = 0;
    uint64_t alg_bytes = 0;
    uint64_t device_bytes = 0;                   // pictures + chunks + words of the original batch: what crossed PCIe
    uint64_t coded_blocks = 0, dense_blocks = 0; // of the original (un-replicated) batch: launch_batch picks the kernel instance by them
    bool any_rgba = false;
    // host copy of what launch_batch needs to keep the RGBA images in step: per picture of the original
    // (un-replicated) batch {stream, cur, MPEGHIP_PIC_RGBA?, covers every macroblock of the frame?}
    struct PicNote { uint32_t stream; uint8_t cur, rgba, full; };
    std::vector<PicNote> notes;
    uint32_t replicas = 1;
};

struct mpeghip_video {
    mpeghip_ctx *ctx = nullptr;
    mpeghip_video_info info{};
    uint8_t *d_frames = nullptr;
    uint8_t *d_rgba = nullptr;
    uint8_t *d_qmat = nullptr;    // [n_streams][64 positions][2 classes]{matrix entry, premultiplier}
    uint64_t *d_hash = nullptr;
    // rgba_sync[stream*3 + slot]: the slot's RGBA image equals the conversion of its planes.  Pictures
    // flagged MPEGHIP_PIC_RGBA convert the macroblocks they write inside the reconstruction kernel; that
    // is the whole story unless a partial picture lands on a slot whose image is out of date — then
    // the whole-frame pass runs as well (launch_batch).
    std::vector<uint8_t> rgba_sync;
    // mpeghip_video_submit: two batches with pinned host staging, used alternately, so that the caller
    // can parse picture N+1 while picture N's copy and kernel are in flight
    struct Staging {
        mpeghip_batch batch;
        uint8_t *h = nullptr;      // pinned
        uint8_t *d_h = nullptr;    // ... as the device sees it (small submits are read in place: upload_into); nullptr: not mapped
        size_t cap_h = 0;
        hipEvent_t done = nullptr; // recorded after the batch's kernel
        hipEvent_t copied = nullptr; // staged commits: recorded on the context's copy stream behind the buffer's H2D copy
        bool in_flight = false;
        // a device-packed stage (mpeghip_video_stage_begin_device): the staged arrays as they are on the device, the packer's
        // scratch, and its verdict — written by pack_gate_kernel into pinned memory, looked at when `done` has passed
        uint8_t *d_raw = nullptr;
        size_t cap_raw = 0;
        uint32_t *d_seen = nullptr;
        size_t cap_seen = 0;
        unsigned long long *d_err = nullptr;     // [pictures of the commit] + the gate's three scratch words
        size_t cap_err = 0;
        unsigned long long *h_verdict = nullptr; // pinned: [0] first report, [1] refused pictures, then their indices (pack_gate_kernel)
        hipEvent_t gated = nullptr;              // recorded behind pack_gate_kernel: the verdict is there (mpeghip_video_verdict)
        bool packed_on_device = false;           // the commit in flight was: h_verdict is meaningful once `gated` has passed
        std::vector<uint32_t> pk_mb_first;       // per picture of that commit: its first macroblock (to name the picture of a report)
        std::vector<uint32_t> pk_stream;         // ... and its stream
        mpeghip_video *owner = nullptr;          // (whose mpeghip_video_refused list a verdict fills)
    } staging[2];
    int next_staging = 0;
    struct mpeghip_stage *stage = nullptr; // the open mpeghip_video_stage_begin, if any
    uint8_t *bounce = nullptr;             // pinned: read_planes / read_rgba land here first
    size_t bounce_cap = 0;
    uint8_t *d_linear = nullptr;           // one slot's planes in the reference's linear layout (read / write_planes)
    // the refused pictures of the verdict last returned (mpeghip_video_refused): (picture index in its commit, stream)
    std::vector<std::pair<uint32_t, uint32_t>> refused;
    uint64_t refused_total = 0;
    // mpeghip_video_read_planes_async: tickets count the read-backs queued; ticket t's event is read_done[t % 4] (events complete
    // in stream order, so a slot that a later read-back has taken over answers for the earlier one too)
    hipEvent_t read_done[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t reads_issued = 0;
    int tile_policy = MPEGHIP_TILE_AUTO;   // mpeghip_video_set_tile_policy
    int n_cu = 256;                        // compute units of the device (asked once, at open)
    // mpeghip_video_host_mirror: every (stream, slot)'s planes once more, LINEAR, in pinned host memory, written by the
    // reconstruction launch itself (recon_wide_kernel<false, true>).  mirror_valid[stream*3 + slot]: the copy equals the slot —
    // kept by launches that mirror, lost to anything else that writes the slot, restored by an untiling launch when asked for.
    uint8_t *h_mirror = nullptr, *d_mirror = nullptr;
    uint64_t mirror_stride = 0;
    std::vector<uint8_t> mirror_valid;
    uint64_t mirror_requests = 0, mirror_repairs = 0; // mpeghip_video_mirror_async calls / those that had to untile the slot first
};

// what validation learns about a picture (the dependency check across pictures needs it)
struct PicUse {
    uint8_t fwd = 0, bwd = 0; // some macroblock predicts from pic.fwd / pic.bwd
};

// mpeghip_video_stage_*: one submit assembled in a staging buffer by several host threads
struct mpeghip_stage {
    mpeghip_video *v = nullptr;
    mpeghip_video::Staging *sg = nullptr;
    uint32_t n_pics = 0, n_mbs = 0, n_chunks = 0;
    std::vector<uint32_t> mb_first, mb_count;   // per picture: its macroblocks [mb_first, mb_first + mb_count)
    std::vector<uint32_t> chunk_first;          // per picture: its first chunk
    std::vector<uint64_t> units;                // per picture: its coefficient bytes
    std::vector<uint64_t> alg;                  // per picture, written by its put
    std::vector<uint32_t> blocks, dense;        // per picture, written by its put: coded blocks / dense units
    std::vector<PicUse> use;                    // per picture, written by its put
    std::unique_ptr<std::atomic<uint8_t>[]> done; // per picture: 0 = not put, 1 = a put has claimed it, 2 = put succeeded
    size_t c_at = 0, w_at = 0;                  // staging layout: pictures | chunks | words
    uint64_t words_cap = 0;                     // room for every picture's worst case
    std::atomic<uint64_t> words_used{0};        // dwords handed out so far: a put packs its picture into scratch
                                                // memory of its thread, then takes exactly the room it needs, so
                                                // that the words form one contiguous block = one H2D copy
    // a device-packed stage: the pinned buffer holds pictures | PkPic | dependency list | macroblocks | words as the caller
    // hands them over; pack_kernel makes chunks and words of them on the device
    bool device = false;
    std::vector<uint32_t> word_first;           // per picture: its first dword in the staged words (a multiple of 16)
    size_t a_at = 0, d_at = 0, m_at = 0, in_at = 0; // staging layout of a device-packed stage
    uint64_t words_total = 0;                   // dwords of the staged words
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
    int n_cu = 256;                // compute units of the device (asked once, at open)
    // mpeghip_audio_synth_async: ticket t's event is synth_done[t % 4] (stream order: a later one answers for an earlier one)
    hipEvent_t synth_done[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t synths_issued = 0;
    bool can_undo = false;         // exactly one launch since the state in the alternate buffers was current (mpeghip_audio_undo_last)
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

// The host NUMA node the context's GPU hangs off (its PCI function's numa_node in sysfs), -1 if unknown: the
// threads that feed the device — parser pool, staged puts, the pinned staging buffers they fill — belong there.
static bool pci_bus_id_of(int device, char (&id)[64])
{
    memset(id, 0, sizeof(id));
    if (hipDeviceGetPCIBusId(id, (int)sizeof(id) - 1, device) != hipSuccess)
        return false;
    for (char *p = id; *p; p++)
        if (*p >= 'A' && *p <= 'F')
            *p = (char)(*p - 'A' + 'a');
    return true;
}
int mpeghip_ctx_pci_bus_id(const mpeghip_ctx *c, char *out, size_t cap)
{
    if (!c || !out)
        return fail(MPEGHIP_ERR_INVALID, "ctx or out is NULL");
    char id[64];
    if (!pci_bus_id_of(c->device, id))
        return fail(MPEGHIP_ERR_HIP, "hipDeviceGetPCIBusId failed for device %d", c->device);
    if (strlen(id) + 1 > cap)
        return fail(MPEGHIP_ERR_INVALID, "out holds %zu bytes, the PCI address needs %zu", cap, strlen(id) + 1);
    memcpy(out, id, strlen(id) + 1);
    return MPEGHIP_OK;
}
int mpeghip_ctx_numa_node(const mpeghip_ctx *c)
{
    if (!c)
        return -1;
    char id[64];
    if (!pci_bus_id_of(c->device, id))
        return -1;
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", id);
    FILE *f = fopen(path, "r");
    if (!f)
        return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1)
        node = -1;
    fclose(f);
    return node;
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
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        if (c->owns_stream)
            (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(MPEGHIP_ERR_HIP, "hipEventCreate / hipStreamCreate failed");
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
    if (c->copy_stream) {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamDestroy(c->copy_stream);
    }
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
    if (width == 0 || height == 0 || width > 4095 || height > 4095 || n_streams == 0 || n_streams > kRcMaxStreams)
        return fail(MPEGHIP_ERR_INVALID, "bad geometry %ux%u x %u streams (at most %u streams per frame store)", width, height, n_streams, kRcMaxStreams);
    HIP_TRY(hipSetDevice(c->device));
    mpeghip_video *v = new (std::nothrow) mpeghip_video();
    if (!v)
        return fail(MPEGHIP_ERR_OOM, "host allocation failed");
    v->ctx = c;
    (void)hipDeviceGetAttribute(&v->n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
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
    // Tail pad behind the last slot: a kRSlow gather always fetches 17 luma rows x 32 bytes / 9 chroma rows x 16 bytes from the dword
    // below the window origin — one row and one piece more than the half-pel mode (and validate_mb) needs — so a window that ends
    // exactly at the end of the slot's pad reads up to luma_w + 47 bytes past frame_bytes (ignored; inside every other slot's stride).
    const uint64_t tail_pad = align_up((uint64_t)in.luma_w + 64, 256);
    const uint64_t total = in.frame_stride * MPEGHIP_SLOTS * n_streams + tail_pad;
    int rc = MPEGHIP_OK;
    do {
        if (hipMalloc((void **)&v->d_frames, total) != hipSuccess) {
            rc = fail(MPEGHIP_ERR_OOM, "hipMalloc(%llu) for the frame store failed", (unsigned long long)total);
            break;
        }
        if (hipMalloc((void **)&v->d_qmat, (size_t)n_streams * kRcQtabStride + kRcQtabPad) != hipSuccess ||
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
            rc_make_qtable(&qm[(size_t)s * 256], k_default_intra, non_intra, k_premult);
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

static void batch_release(mpeghip_batch *b)
{
    if (b->d_blob)
        (void)hipFree(b->d_blob);
    b->d_blob = nullptr;
    b->cap_blob = 0;
    b->d_pics = nullptr;
    b->d_chunks = b->d_words = nullptr;
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
    if (v->d_linear)
        (void)hipFree(v->d_linear);
    if (v->h_mirror)
        (void)hipHostFree(v->h_mirror);
    for (auto &e : v->read_done)
        if (e)
            (void)hipEventDestroy(e);
    for (auto &sg : v->staging) {
        batch_release(&sg.batch);
        if (sg.d_raw)
            (void)hipFree(sg.d_raw);
        if (sg.d_seen)
            (void)hipFree(sg.d_seen);
        if (sg.d_err)
            (void)hipFree(sg.d_err);
        if (sg.h_verdict)
            (void)hipHostFree(sg.h_verdict);
        if (sg.gated)
            (void)hipEventDestroy(sg.gated);
        if (sg.h)
            (void)hipHostFree(sg.h);
        if (sg.done)
            (void)hipEventDestroy(sg.done);
        if (sg.copied)
            (void)hipEventDestroy(sg.copied);
    }
    if (v->d_frames)
        (void)hipFree(v->d_frames);
    if (v->d_rgba)
        (void)hipFree(v->d_rgba);
    if (v->d_qmat)
        (void)hipFree(v->d_qmat);
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
    rc_make_qtable(t, intra, non_intra, k_premult);
    HIP_TRY(hipSetDevice(v->ctx->device));
    HIP_TRY(hipStreamSynchronize(v->ctx->stream)); // earlier pictures may still read the old matrices
    HIP_TRY(hipMemcpy(v->d_qmat + (size_t)stream * 256, t, 256, hipMemcpyHostToDevice));
    return MPEGHIP_OK;
}

int mpeghip_video_set_tile_policy(mpeghip_video *v, int policy)
{
    if (!v || policy < MPEGHIP_TILE_AUTO || policy > MPEGHIP_TILE_INT32)
        return fail(MPEGHIP_ERR_INVALID, "bad argument");
    v->tile_policy = policy;
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

static uint64_t rgba_stride_of(const mpeghip_video *v) { return align_up(v->info.rgba_bytes, 256); }

// ---- host-side validation of a submit; also totals the algorithmic bytes
static int validate_pic(const mpeghip_video_info &in, const mpeghip_pic_desc &pd, uint32_t p)
{
    if (pd.stream >= in.n_streams || pd.cur >= MPEGHIP_SLOTS || pd.fwd >= MPEGHIP_SLOTS || pd.bwd >= MPEGHIP_SLOTS)
        return fail(MPEGHIP_ERR_INVALID, "picture %u: bad stream/slot", p);
    if (pd.flags & ~(MPEGHIP_PIC_RGBA | MPEGHIP_PIC_SPARSE)) // (a stray bit from an uninitialised caller must not pick the coefficient form)
        return fail(MPEGHIP_ERR_INVALID, "picture %u: flags 0x%x: undefined bits", p, pd.flags);
    return MPEGHIP_OK;
}

// One macroblock of picture `pd`: field ranges, coefficient extent inside [0, coef_units), prediction
// reads inside the frame buffer.  Adds its algorithmic bytes to *alg.
static int validate_mb(const mpeghip_video_info &in, const mpeghip_pic_desc &pd, const mpeghip_mb_desc &m, uint32_t i,
                       uint64_t coef_units, uint64_t *alg, uint64_t *named_units)
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
    *named_units += units;
    if (!raw && nb && (m.qscale == 0 || m.qscale > 31))
        return fail(MPEGHIP_ERR_INVALID, "macroblock %u: quantiser_scale %u", i, m.qscale);
    uint64_t ref_bytes = 0;
    if (!intra) {
        const uint8_t ref = (m.flags & MPEGHIP_MB_REF_BWD) ? pd.bwd : pd.fwd;
        if (ref == pd.cur)
            return fail(MPEGHIP_ERR_INVALID, "macroblock %u: predicts from the slot its picture writes (%u)", i, ref);
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
    *alg += 32 + units * MPEGHIP_COEF_UNIT + ref_bytes + (intra ? 64ull * nb : 384);
    if (pd.flags & MPEGHIP_PIC_RGBA)
        *alg += 1024;
    return MPEGHIP_OK;
}

// All macroblocks of ONE picture.  Macroblocks of one submit run concurrently, so a position may be named once
// only (the reference lets a damaged stream address a macroblock twice, video.go:462-486: the emitter starts a
// new submit there); `seen` is scratch of at least mb_w * mb_h bits.
// *named_units: the coefficient units the blocks validated so far name, counted with repetition.  The device-format
// buffers are sized from the coefficient buffer (rc_max_words), so blocks may share units only as far as the buffer has
// as many units as blocks name: beyond that the submit is refused instead of overrunning them.
static int validate_picture(const mpeghip_video_info &in, const mpeghip_pic_desc &pd, uint32_t p, const mpeghip_mb_desc *mbs,
                            uint32_t n, uint32_t first_index, uint64_t coef_units, uint64_t *alg, PicUse *use,
                            std::vector<uint64_t> &seen, uint64_t *named_units)
{
    seen.assign(((size_t)in.mb_w * in.mb_h + 63) / 64, 0);
    PicUse u;
    for (uint32_t k = 0; k < n; k++) {
        const mpeghip_mb_desc &m = mbs[k];
        const int rc = validate_mb(in, pd, m, first_index + k, coef_units, alg, named_units);
        if (rc != MPEGHIP_OK)
            return rc;
        if (*named_units > coef_units)
            return fail(MPEGHIP_ERR_INVALID, "macroblock %u: the coded blocks up to here name %llu coefficient units, the buffer "
                        "holds %llu (blocks may not share units beyond that)", first_index + k, (unsigned long long)*named_units,
                        (unsigned long long)coef_units);
        const uint32_t at = (uint32_t)m.mb_y * in.mb_w + m.mb_x;
        if (seen[at >> 6] & (1ull << (at & 63)))
            return fail(MPEGHIP_ERR_INVALID, "picture %u: macroblock (%u,%u) is addressed twice in one submit", p, m.mb_x, m.mb_y);
        seen[at >> 6] |= 1ull << (at & 63);
        u.fwd |= (m.flags & MPEGHIP_MB_REF_FWD) ? 1 : 0;
        u.bwd |= (m.flags & MPEGHIP_MB_REF_BWD) ? 1 : 0;
    }
    *use = u;
    return MPEGHIP_OK;
}

// Pictures of one stream inside one submit run concurrently too: none may write a slot another one writes or
// predicts from.
static int check_dependencies(const mpeghip_pic_desc *pics, const PicUse *use, uint32_t n_pics)
{
    if (n_pics < 2)
        return MPEGHIP_OK;
    std::vector<uint32_t> order(n_pics);
    for (uint32_t p = 0; p < n_pics; p++)
        order[p] = p;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        return pics[x].stream != pics[y].stream ? pics[x].stream < pics[y].stream : x < y;
    });
    for (uint32_t i = 0; i < n_pics;) {
        uint32_t j = i + 1;
        while (j < n_pics && pics[order[j]].stream == pics[order[i]].stream)
            j++;
        for (uint32_t x = i; x < j; x++)
            for (uint32_t y = i; y < j; y++) {
                if (x == y)
                    continue;
                const mpeghip_pic_desc &w = pics[order[x]], &r = pics[order[y]];
                if ((x < y && w.cur == r.cur) || (use[order[y]].fwd && r.fwd == w.cur) || (use[order[y]].bwd && r.bwd == w.cur))
                    return fail(MPEGHIP_ERR_INVALID, "pictures %u and %u of stream %u depend on each other (slot %u): they need "
                                "separate submits", order[x], order[y], w.stream, w.cur);
            }
        i = j;
    }
    return MPEGHIP_OK;
}

static RcGeom record_geometry(const mpeghip_video *v)
{
    RcGeom g;
    g.mb_w = v->info.mb_w;
    g.mb_h = v->info.mb_h;
    g.luma_w = v->info.luma_w;
    g.chroma_w = v->info.chroma_w;
    g.luma_bytes = (uint32_t)v->info.luma_bytes;
    g.frame_stride = v->info.frame_stride;
    g.rgba_stride = rgba_stride_of(v);
    return g;
}

// chunks a submit needs: every picture's macroblocks are padded to whole chunks
static uint64_t chunks_of(const mpeghip_pic_desc *pics, uint32_t n_pics)
{
    uint64_t n = 0;
    for (uint32_t p = 0; p < n_pics; p++)
        n += rc_max_chunks(pics[p].mb_count);
    return n;
}

// room the packed form of a picture can need (dwords), whichever form it arrives in: a unit becomes at most 65 words, a
// sparse picture at most its own dwords + a block word per snapshot block
static size_t words_room(uint64_t coef_bytes, uint64_t n_mbs)
{
    const size_t as_units = rc_max_words((coef_bytes + MPEGHIP_COEF_UNIT - 1) / MPEGHIP_COEF_UNIT);
    const size_t as_sparse = rc_max_words_sparse(coef_bytes / 4, (uint32_t)(n_mbs > 0xffffffffull ? 0xffffffffull : n_mbs));
    return as_units > as_sparse ? as_units : as_sparse;
}
// ... of a submit whose pictures may come in both forms: unit pictures can need 65 words per unit of the buffer, sparse
// pictures the buffer's dwords once more
static size_t words_room_submit(const mpeghip_pic_desc *pics, uint32_t n_pics, uint64_t coef_bytes, uint64_t n_mbs)
{
    bool units = false, sparse = false;
    for (uint32_t p = 0; p < n_pics; p++)
        ((pics[p].flags & MPEGHIP_PIC_SPARSE) ? sparse : units) = true;
    if (units && sparse)
        return rc_max_words((coef_bytes + MPEGHIP_COEF_UNIT - 1) / MPEGHIP_COEF_UNIT) + rc_max_words_sparse(coef_bytes / 4, 0);
    return words_room(coef_bytes, n_mbs);
}
static std::string sparse_error_text(uint32_t pic, uint32_t mb)
{
    char t[480];
    snprintf(t, sizeof(t), "picture %u, macroblock %u: malformed sparse block data (a count beyond 64 — other than 64 for a snapshot "
             "block —, a block that ends behind the coefficient buffer, a macroblock whose data begins before the previous one's "
             "ends, bits outside a pair's two fields, or an intra block without its DC first)", pic, mb);
    return t;
}

// Validate a whole submit and (chunks_out != NULL) pack it into the device format.  The macroblocks of picture p
// are mbs[mb_first .. mb_first + mb_count); every macroblock belongs to exactly one picture's range.
static int validate_and_pack(const mpeghip_video *v, const mpeghip_pic_desc *pics, uint32_t n_pics,
                             const mpeghip_mb_desc *mbs, uint32_t n_mbs, const void *coefs, size_t coef_bytes,
                             uint64_t *alg_bytes, uint32_t *chunks_out, uint32_t *words_out, uint64_t *n_words,
                             uint64_t *stats = nullptr /* -> {coded blocks, dense units} */)
{
    const mpeghip_video_info &in = v->info;
    if (n_pics && !pics)
        return fail(MPEGHIP_ERR_INVALID, "pics is NULL");
    if (n_mbs && !mbs)
        return fail(MPEGHIP_ERR_INVALID, "mbs is NULL");
    if (n_mbs > 0 && coef_bytes > 0 && !coefs)
        return fail(MPEGHIP_ERR_INVALID, "coefs is NULL");
    bool any_units = false;
    for (uint32_t p = 0; p < n_pics; p++)
        any_units = any_units || !(pics[p].flags & MPEGHIP_PIC_SPARSE);
    if (coef_bytes % (any_units ? MPEGHIP_COEF_UNIT : 4))
        return fail(MPEGHIP_ERR_INVALID, "coef_bytes %zu is not a multiple of %d", coef_bytes, any_units ? MPEGHIP_COEF_UNIT : 4);
    uint64_t covered = 0;
    for (uint32_t p = 0; p < n_pics; p++) {
        const int rc = validate_pic(in, pics[p], p);
        if (rc != MPEGHIP_OK)
            return rc;
        if ((uint64_t)pics[p].mb_first + pics[p].mb_count > n_mbs)
            return fail(MPEGHIP_ERR_INVALID, "picture %u: macroblock range out of bounds", p);
        covered += pics[p].mb_count;
    }
    if (covered != n_mbs)
        return fail(MPEGHIP_ERR_INVALID, "the pictures' macroblock ranges cover %llu of %u macroblocks",
                    (unsigned long long)covered, n_mbs);
    const RcGeom geom = record_geometry(v);
    const uint64_t coef_units = coef_bytes / MPEGHIP_COEF_UNIT;
    std::vector<PicUse> use(n_pics);
    std::vector<uint64_t> seen;
    uint64_t alg = 0, words = 0, chunk = 0, named_units = 0, blocks = 0, dense = 0, sparse_used = 0;
    for (uint32_t p = 0; p < n_pics; p++) {
        const mpeghip_mb_desc *pm = mbs + pics[p].mb_first;
        for (uint32_t k = 0; k < pics[p].mb_count; k++)
            if (pm[k].pic != p)
                return fail(MPEGHIP_ERR_INVALID, "macroblock %u: names picture %u but lies in picture %u's range",
                            pics[p].mb_first + k, pm[k].pic, p);
        const bool sparse = (pics[p].flags & MPEGHIP_PIC_SPARSE) != 0;
        uint64_t ignored = 0; // (sparse: the coefficient extents are in the words themselves: the packer checks them)
        const int rc = validate_picture(in, pics[p], p, pm, pics[p].mb_count, pics[p].mb_first, sparse ? ~0ull >> 2 : coef_units, &alg,
                                        &use[p], seen, sparse ? &ignored : &named_units);
        if (rc != MPEGHIP_OK)
            return rc;
        if (chunks_out) {
            if (words > 0xffffffffull - words_room_submit(pics, n_pics, coef_bytes, n_mbs))
                return fail(MPEGHIP_ERR_INVALID, "batch too large for 32-bit indices");
            const RcPacked got = sparse ? rc_pack_picture<true, true>(geom, pics[p], pm, pics[p].mb_count, static_cast<const uint8_t *>(coefs),
                                                                      (uint32_t)words, chunks_out + chunk * kRcChunkDwords, words_out + words,
                                                                      coef_bytes / 4, rc_max_words_sparse(coef_bytes / 4, 0) - sparse_used)
                                        : rc_pack_picture(geom, pics[p], pm, pics[p].mb_count, static_cast<const uint8_t *>(coefs),
                                                 (uint32_t)words, chunks_out + chunk * kRcChunkDwords, words_out + words);
            if (got.bad)
                return fail(MPEGHIP_ERR_INVALID, "%s", sparse_error_text(p, pics[p].mb_first + got.bad - 1).c_str());
            chunk += got.chunks;
            words += got.words;
            sparse_used += sparse ? got.words : 0; // (sparse pictures together: no more than the buffer's dwords — pictures may
                                                   // name the same words, but not past the room the buffers were sized for)
            blocks += got.blocks;
            dense += got.dense_blocks;
        }
    }
    if (stats) {
        stats[0] = blocks;
        stats[1] = dense;
    }
    const int rc = check_dependencies(pics, use.data(), n_pics);
    if (rc != MPEGHIP_OK)
        return rc;
    if (alg_bytes)
        *alg_bytes = alg;
    if (n_words)
        *n_words = words;
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

static int grow_pinned(mpeghip_video::Staging *sg, size_t need)
{
    if (need <= sg->cap_h)
        return MPEGHIP_OK;
    if (sg->h)
        (void)hipHostFree(sg->h);
    sg->h = sg->d_h = nullptr;
    sg->cap_h = 0;
    const size_t cap = need + need / 2;
    HIP_TRY(hipHostMalloc((void **)&sg->h, cap, hipHostMallocDefault));
    sg->cap_h = cap;
    void *d = nullptr;
    sg->d_h = hipHostGetDevicePointer(&d, sg->h, 0) == hipSuccess ? static_cast<uint8_t *>(d) : nullptr;
    (void)hipGetLastError();
    return MPEGHIP_OK;
}

// ---- the verdict of a device-packed commit (mpeghip_video_stage_begin_device): known once the staging slot's `gated` event has
// passed.  Reasons are those of validate_mb / rc_pack_picture (video_pack_lane.h: kPk*).  A refusal is PER PICTURE (round 6): the
// commit's other pictures — other streams — were reconstructed; which pictures were refused: mpeghip_video_refused.
static int reap_verdict(mpeghip_video::Staging *sg)
{
    if (!sg->packed_on_device)
        return MPEGHIP_OK;
    sg->packed_on_device = false;
    const volatile unsigned long long *hv = sg->h_verdict;
    const unsigned long long key = hv[0];
    if (key == kPkNoError)
        return MPEGHIP_OK;
    const unsigned long long n_refused = hv[1];
    const uint32_t mb = (uint32_t)(key >> 8), reason = (uint32_t)(key & 0xff);
    uint32_t pic = 0;
    if (!sg->pk_mb_first.empty())
        pic = (uint32_t)(std::upper_bound(sg->pk_mb_first.begin(), sg->pk_mb_first.end(), mb) - sg->pk_mb_first.begin()) - 1;
    const uint32_t k = sg->pk_mb_first.empty() ? mb : mb - sg->pk_mb_first[pic];
    const uint32_t stream = pic < sg->pk_stream.size() ? sg->pk_stream[pic] : 0;
    if (sg->owner) { // which pictures (mpeghip_video_refused): the gate's list, in picture order
        auto &out = sg->owner->refused;
        out.clear();
        const volatile uint32_t *list = reinterpret_cast<const volatile uint32_t *>(hv + 2);
        for (unsigned long long i = 0; i < n_refused && i < kPkVerdictList; i++) {
            const uint32_t p = list[i];
            out.emplace_back(p, p < sg->pk_stream.size() ? sg->pk_stream[p] : 0u);
        }
        std::sort(out.begin(), out.end());
        sg->owner->refused_total = n_refused;
    }
    static const char *const why[] = {
        "", "position outside the picture", "flags name no or two references (or one for an intra macroblock)", "cbp beyond 0x3f",
        "quantiser_scale outside 1..31", "predicts from the slot its picture writes", "motion vector reads outside the frame buffer",
        "its position is addressed twice in one picture", "malformed sparse block data (a count beyond 64 — other than 64 for a snapshot "
        "block —, a block that ends behind the picture's words, bits outside a pair's two fields, or an intra block without its DC first)",
        "malformed sparse block data: its data begins before the previous macroblock's ends (macroblocks name their words in order)",
        "its picture and another picture of the same stream in this commit depend on each other: they need separate commits"};
    return fail(reason == kPkRange ? MPEGHIP_ERR_RANGE : MPEGHIP_ERR_INVALID,
                "device-packed commit: %llu of its %zu pictures refused and not reconstructed (the others were): picture %u (stream %u), "
                "macroblock %u: %s", n_refused, sg->pk_mb_first.size(), pic, stream, k, reason < sizeof(why) / sizeof(why[0]) ? why[reason] : "?");
}
// a staging slot is about to be reused, or the caller waits for the device: its last commit has finished
static int retire(mpeghip_video::Staging *sg)
{
    if (sg->in_flight) {
        HIP_TRY(hipEventSynchronize(sg->done));
        sg->in_flight = false;
    }
    return reap_verdict(sg);
}
// everything queued on this handle has finished (the caller has synchronised the stream): deferred verdicts, oldest first
static int reap_all(mpeghip_video *v)
{
    const int older = reap_verdict(&v->staging[v->next_staging]);
    if (older != MPEGHIP_OK) // (the newer commit's verdict, if it has one, is the next synchronising call's)
        return older;
    return reap_verdict(&v->staging[v->next_staging ^ 1]);
}

// Which instance of the reconstruction kernel a batch runs on: the one built for dense units (kT16 = false) when more than ONE
// EIGHTH of its coded blocks are dense units.  Measured (profiles/round5_p_dense_share_crossover_lds_transposition.txt: 256 1080p
// streams at their own GOP phases, a share of them with dense content, both instances interleaved on one box): dense block share
// 0.01 / 0.24 / 0.33 / 0.41 / 0.56 / 0.68 / 1.00 -> tie (+0.3 %) / +3.0 % / +4.3 % / +4.9 % / +8.0 % / +8.3 % / +13.2 % for that
// instance: it is never behind, and typical batches (share 0.01) stay on the instance the headline was measured on.  (Rounds 2 - 5
// ran dense batches on an int32-tile instance with 7 waves per SIMD, which lost to the DPP instance below a third: round5_l_*.)
// A device-packed commit, whose blocks the host has not looked at, goes by its input dwords per macroblock: the same sweep's
// 24 / 70 / 93 / 116 / 162 / 207 / 390 (an eighth: 48).
constexpr uint64_t kDenseWordsPerMb = 48;
constexpr uint64_t kDenseShareNum = 1, kDenseShareDen = 8;
constexpr int kReconWaves = 1; // waves (= chunks) per workgroup: waves of a workgroup that finish early keep their slots until the
                               // last one has (its LDS goes back as a whole) — 1 beats 2 beats 4 (profiles/r2w_ab_waves_per_workgroup.txt)

// Behind a reconstruction launch, Frame.RGBA bookkeeping: the kernel has converted every macroblock that flagged pictures wrote.
static int finish_rgba_bookkeeping(mpeghip_video *v, const mpeghip_batch *b, const VideoArgs &a)
{
    const mpeghip_video_info &in = v->info;
    hipStream_t st = v->ctx->stream;
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
        for (uint64_t p0 = 0; p0 < b->n_pics; p0 += 32768) {
            const uint32_t np = (uint32_t)(b->n_pics - p0 < 32768 ? b->n_pics - p0 : 32768);
            hipLaunchKernelGGL(rgba_pics_kernel, dim3((in.mb_w + 7) / 8, in.mb_h, np), dim3(256), 0, st, a,
                               (uint32_t)p0);
        }
        HIP_TRY(hipGetLastError());
    }
    return MPEGHIP_OK;
}

// the slots a batch writes are no longer what their host mirrors hold (a launch without the mirror code wrote them)
static void mirror_lost(mpeghip_video *v, const mpeghip_batch *b)
{
    if (!v->h_mirror)
        return;
    for (uint32_t r = 0; r < b->replicas; r++)
        for (const mpeghip_batch::PicNote &n : b->notes)
            v->mirror_valid[((size_t)n.stream + r) * MPEGHIP_SLOTS + n.cur] = 0;
}

static int launch_batch(mpeghip_video *v, const mpeghip_batch *b)
{
    if (b->n_chunks == 0)
        return MPEGHIP_OK;
    const mpeghip_video_info &in = v->info;
    VideoArgs a;
    if (b->n_chunks >= (1ull << 25))
        return fail(MPEGHIP_ERR_INVALID, "batch too large for one launch: %llu chunks (a chunk's byte offset is a 32-bit product)",
                    (unsigned long long)b->n_chunks);
    a.frames = v->d_frames;
    a.frames_b = v->d_frames - kRcDmaBias;
    a.frame_stride = in.frame_stride;
    a.mb_w = in.mb_w;
    a.mb_h = in.mb_h;
    a.luma_w = in.luma_w;
    a.luma_h = in.luma_h;
    a.chroma_w = in.chroma_w;
    a.chroma_h = in.chroma_h;
    a.luma_bytes = (uint32_t)in.luma_bytes;
    a.chroma_bytes = (uint32_t)in.chroma_bytes;
    a.pics = b->d_pics;
    a.chunks = b->d_chunks;
    a.words = b->d_words;
    a.qmat = v->d_qmat;
    a.n_chunks = (uint32_t)b->n_chunks;
    a.width = in.width;
    a.height = in.height;
    a.rgba = v->d_rgba;
    a.rgba_stride = rgba_stride_of(v);
    hipStream_t st = v->ctx->stream;
    // Which instance (video_recon_lane.h, "the wave's coefficient tile"): batches with dense units are bound by vector-ALU issue
    // and want the transposition through LDS and the short dequantisation; the others the transposition by DPP (no LDS round
    // trip), with Frame.RGBA fused as well.  The switch-over: kDenseShareNum / kDenseShareDen above.
    bool t16 = b->dense_blocks * kDenseShareDen <= b->coded_blocks * kDenseShareNum;
    if (v->tile_policy != MPEGHIP_TILE_AUTO)
        t16 = v->tile_policy == MPEGHIP_TILE_INT16;
    // A launch that leaves most slots empty goes to recon_wide_kernel, four waves per chunk: its duration is one chunk's chain, which
    // four waves walk in parallel.  Measured against recon_kernel for launches of 1 .. 16 1080p pictures
    // (profiles/round5_k_ab_wide_kernel_by_launch_size.txt): ahead for one and two pictures (-27 % / -9 .. -15 % typical, -25 % /
    // -3 .. -5 % dense), behind from three on — so up to twice the device's wave slots.  (Only when the library picks: a pinned
    // policy names recon_kernel's instances.)
    if (v->tile_policy == MPEGHIP_TILE_AUTO && a.n_chunks * 4 <= (uint64_t)(v->n_cu > 0 ? v->n_cu : 256) * 4 * 8 * 2) {
        const uint32_t g8 = (a.n_chunks + 7) / 8;
        if (b->any_rgba)
            hipLaunchKernelGGL((recon_wide_kernel<true, false>), dim3(g8 * 8), dim3(256), 0, st, g8, a.n_chunks, a.chunks, a.words, a.qmat,
                               a.frames_b, a.mb_w, a.luma_bytes, a.rgba, a.rgba_stride, a.width, a.height);
        else if (v->d_mirror) // a store with a host mirror: the launch writes the frames' linear copies too (they stay valid)
            hipLaunchKernelGGL((recon_wide_kernel<false, true>), dim3(g8 * 8), dim3(256), 0, st, g8, a.n_chunks, a.chunks, a.words, a.qmat,
                               a.frames_b, a.mb_w, a.luma_bytes, v->d_mirror, v->mirror_stride, a.width, a.height);
        else
            hipLaunchKernelGGL((recon_wide_kernel<false, false>), dim3(g8 * 8), dim3(256), 0, st, g8, a.n_chunks, a.chunks, a.words, a.qmat,
                               a.frames_b, a.mb_w, a.luma_bytes, a.rgba, a.rgba_stride, a.width, a.height);
        HIP_TRY(hipGetLastError());
        if (b->any_rgba)
            mirror_lost(v, b);
        return finish_rgba_bookkeeping(v, b, a);
    }
    mirror_lost(v, b); // (recon_kernel carries no mirror code: launches that fill the device are not a lone decoder's)
    // One chunk per wave (the comment at recon_kernel).  A multiple of 8 workgroups: the kernel's XCD remap is then a multiply-add;
    // the at most 7 surplus waves return at once.
    const uint32_t grid8 = ((a.n_chunks + kReconWaves - 1) / kReconWaves + 7) / 8;
    const uint32_t grid = grid8 * 8;
#ifdef MPG_PROBE_PAIRS
    a.rgba_stride = (uint64_t)(in.n_streams / 2) * MPEGHIP_SLOTS * in.frame_stride; // (the probe's read bias; no RGBA in its runs)
#endif
#define LAUNCH_RECON(RGBA, T16) \
    hipLaunchKernelGGL((recon_kernel<kReconWaves, RGBA, T16>), dim3(grid), dim3(kReconWaves * 64), 0, st, grid8, a.n_chunks, a.chunks, a.words, \
                       a.qmat, a.frames_b, a.mb_w, a.luma_bytes, a.rgba, a.rgba_stride, a.width, a.height)
    if (t16) {
        if (b->any_rgba)
            LAUNCH_RECON(true, true);
        else
            LAUNCH_RECON(false, true);
    } else {
        if (b->any_rgba)
            LAUNCH_RECON(true, false);
        else
            LAUNCH_RECON(false, false);
    }
#undef LAUNCH_RECON
    HIP_TRY(hipGetLastError());
    return finish_rgba_bookkeeping(v, b, a);
}

static bool wants_rgba(const mpeghip_pic_desc *pics, uint32_t n_pics)
{
    for (uint32_t p = 0; p < n_pics; p++)
        if (pics[p].flags & MPEGHIP_PIC_RGBA)
            return true;
    return false;
}

static void fill_notes(const mpeghip_video *v, mpeghip_batch *b, const mpeghip_pic_desc *pics, uint32_t n_pics)
{
    b->notes.resize(n_pics);
    for (uint32_t p = 0; p < n_pics; p++) {
        b->notes[p].stream = pics[p].stream;
        b->notes[p].cur = pics[p].cur;
        b->notes[p].rgba = (pics[p].flags & MPEGHIP_PIC_RGBA) ? 1 : 0;
        // (validation has seen every position at most once: as many macroblocks as the frame has = all of them)
        b->notes[p].full = pics[p].mb_count == v->info.mb_w * v->info.mb_h ? 1 : 0;
    }
}

// region offsets of a batch image: pictures | chunks | words
struct BlobLayout {
    size_t c_at, w_at;
};
static BlobLayout blob_layout(uint64_t n_pics, uint64_t n_chunks)
{
    BlobLayout l;
    l.c_at = (sizeof(mpeghip_pic_desc) * (size_t)n_pics + 127) & ~(size_t)127; // (a chunk is one 128-byte line)
    l.w_at = (l.c_at + (size_t)n_chunks * kRcChunkDwords * 4 + 63) & ~(size_t)63;
    return l;
}

// `sg` != nullptr: b is that staging slot's batch; the submit is packed into its pinned buffer and the call
// returns with the copy still in flight.  Otherwise (resident batches) the image is packed in pageable memory,
// copied, replicated on the device, and the call waits.
// submits out of pinned staging up to this size are read by the kernel in place (upload_into); MPEGHIP_ZERO_COPY_SUBMIT=0 in the
// environment switches it off (A/B runs)
static const size_t kZeroCopySubmitBytes = [] {
    const char *e = getenv("MPEGHIP_ZERO_COPY_SUBMIT");
    return e && e[0] == '0' ? (size_t)0 : (size_t)(e && atol(e) > 1 ? atol(e) : 64 << 10);
}();
static int upload_into(mpeghip_video *v, mpeghip_batch *b, const mpeghip_pic_desc *pics, uint32_t n_pics,
                       const mpeghip_mb_desc *mbs, uint32_t n_mbs, const void *coefs, size_t coef_bytes,
                       uint32_t replicas, mpeghip_video::Staging *sg = nullptr)
{
    if ((uint64_t)n_mbs * replicas > 0xffffffffull)
        return fail(MPEGHIP_ERR_INVALID, "batch too large for 32-bit indices");
    if (n_pics && !pics)
        return fail(MPEGHIP_ERR_INVALID, "pics is NULL");
    uint64_t covered = 0;
    for (uint32_t p = 0; p < n_pics; p++) {
        if ((uint64_t)pics[p].mb_first + pics[p].mb_count > n_mbs)
            return fail(MPEGHIP_ERR_INVALID, "picture %u: macroblock range out of bounds", p);
        covered += pics[p].mb_count;
    }
    if (covered != n_mbs) // (before anything is sized from the pictures' counts: overlapping ranges would ask for a huge buffer)
        return fail(MPEGHIP_ERR_INVALID, "the pictures' macroblock ranges cover %llu of %u macroblocks", (unsigned long long)covered, n_mbs);
    HIP_TRY(hipSetDevice(v->ctx->device));
    const uint64_t n_chunks = chunks_of(pics, n_pics);
    const BlobLayout l = blob_layout(n_pics, n_chunks);
    const size_t words_cap = words_room_submit(pics, n_pics, coef_bytes, n_mbs) + kRcWordsPad;
    const size_t pb = sizeof(mpeghip_pic_desc) * (size_t)n_pics;
    hipStream_t st = v->ctx->stream;
    std::vector<uint8_t> pageable;
    uint8_t *h;
    int rc;
    if (sg) {
        if ((rc = retire(sg)) != MPEGHIP_OK) // two submits ago: normally long finished (a device-packed commit's verdict: here)
            return rc;
        if ((rc = grow_pinned(sg, l.w_at + words_cap * 4 + 64)) != MPEGHIP_OK)
            return rc;
        if (!sg->done)
            HIP_TRY(hipEventCreateWithFlags(&sg->done, hipEventDisableTiming));
        h = sg->h;
    } else {
        try {
            pageable.resize(l.w_at + words_cap * 4 + 64);
        } catch (const std::bad_alloc &) { // (nothing throws across the C ABI)
            return fail(MPEGHIP_ERR_OOM, "host allocation of %zu bytes failed", l.w_at + words_cap * 4 + 64);
        }
        h = pageable.data();
    }
    uint64_t n_words = 0;
    uint64_t stats[2] = {0, 0};
    rc = validate_and_pack(v, pics, n_pics, mbs, n_mbs, coefs, coef_bytes, &b->alg_bytes,
                           reinterpret_cast<uint32_t *>(h + l.c_at), reinterpret_cast<uint32_t *>(h + l.w_at), &n_words, stats);
    if (rc != MPEGHIP_OK)
        return rc;
    b->coded_blocks = stats[0];
    b->dense_blocks = stats[1];
    if (n_words * replicas > 0xffffffffull - kRcWordsPad)
        return fail(MPEGHIP_ERR_INVALID, "batch too large for 32-bit indices");
    b->any_rgba = wants_rgba(pics, n_pics);
    if (b->any_rgba && (rc = ensure_rgba(v)) != MPEGHIP_OK)
        return rc;
    if (pb)
        memcpy(h, pics, pb);
    if (replicas == 1) {
        // the device image is the host image: one copy — or, for a SMALL submit out of pinned staging (a lone decoder's picture:
        // 160x120 is 3 KB, SIF 15 KB), none: the kernel reads the chunks and their words from the pinned buffer itself (the device
        // sees pinned host memory; every byte is read once).  The copy call costs the host 3 - 4 us, more than the picture's
        // parse is worth at that size; the staging slot stays untouched until its `done` event has passed either way (retire).
        const size_t total = l.w_at + (size_t)n_words * 4;
        if (sg && sg->d_h && total && total <= kZeroCopySubmitBytes) {
            uint8_t *d = sg->d_h;
            b->d_pics = reinterpret_cast<mpeghip_pic_desc *>(d);
            b->d_chunks = reinterpret_cast<uint32_t *>(d + l.c_at);
            b->d_words = reinterpret_cast<uint32_t *>(d + l.w_at);
        } else {
            if ((rc = grow((void **)&b->d_blob, &b->cap_blob, total + kRcWordsPad * 4)) != 0)
                return rc;
            b->d_pics = reinterpret_cast<mpeghip_pic_desc *>(b->d_blob);
            b->d_chunks = reinterpret_cast<uint32_t *>(b->d_blob + l.c_at);
            b->d_words = reinterpret_cast<uint32_t *>(b->d_blob + l.w_at);
            if (total)
                HIP_TRY(hipMemcpyAsync(b->d_blob, h, total, hipMemcpyHostToDevice, st));
        }
    } else {
        const BlobLayout lr = blob_layout((uint64_t)n_pics * replicas, n_chunks * replicas);
        const size_t total = lr.w_at + (size_t)n_words * 4 * replicas;
        if ((rc = grow((void **)&b->d_blob, &b->cap_blob, total + kRcWordsPad * 4)) != 0)
            return rc;
        b->d_pics = reinterpret_cast<mpeghip_pic_desc *>(b->d_blob);
        b->d_chunks = reinterpret_cast<uint32_t *>(b->d_blob + lr.c_at);
        b->d_words = reinterpret_cast<uint32_t *>(b->d_blob + lr.w_at);
        if (pb)
            HIP_TRY(hipMemcpyAsync(b->d_pics, h, pb, hipMemcpyHostToDevice, st));
        if (n_chunks)
            HIP_TRY(hipMemcpyAsync(b->d_chunks, h + l.c_at, (size_t)n_chunks * kRcChunkDwords * 4, hipMemcpyHostToDevice, st));
        if (n_words)
            HIP_TRY(hipMemcpyAsync(b->d_words, h + l.w_at, (size_t)n_words * 4, hipMemcpyHostToDevice, st));
        for (uint32_t s = 1; s < replicas && n_words; s++)
            HIP_TRY(hipMemcpyAsync(b->d_words + (size_t)s * n_words, b->d_words, (size_t)n_words * 4, hipMemcpyDeviceToDevice, st));
        const uint64_t work = (n_chunks > n_pics ? n_chunks : (uint64_t)n_pics) * replicas;
        ReplicateSteps k;
        k.words = (uint32_t)n_words;
        k.frames = (uint64_t)MPEGHIP_SLOTS * v->info.frame_stride;
        if (work) {
            hipLaunchKernelGGL(replicate_kernel, dim3((uint32_t)((work + 255) / 256)), dim3(256), 0, st, b->d_pics, n_pics,
                               b->d_chunks, (uint32_t)n_chunks, n_mbs, k, replicas);
            HIP_TRY(hipGetLastError());
        }
    }
    if (!sg) // pageable host memory: the copies above may still be reading it
        HIP_TRY(hipStreamSynchronize(st));
    fill_notes(v, b, pics, n_pics);
    b->device_bytes = l.w_at + n_words * 4;
    b->replicas = replicas;
    b->n_pics = (uint64_t)n_pics * replicas;
    b->n_mbs = (uint64_t)n_mbs * replicas;
    b->n_chunks = n_chunks * replicas;
    b->alg_bytes *= replicas;
    return MPEGHIP_OK;
}

static int mpeghip_video_submit_impl(mpeghip_video *v, const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs,
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

int mpeghip_video_submit(mpeghip_video *v, const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs,
                         uint32_t n_mbs, const void *coefs, size_t coef_bytes)
{
    return no_throw([&] { return mpeghip_video_submit_impl(v, pics, n_pics, mbs, n_mbs, coefs, coef_bytes); });
}

static int mpeghip_video_stage_begin_impl(mpeghip_video *v, uint32_t n_pics, const uint32_t *n_mbs, const size_t *coef_bytes,
                              mpeghip_stage **out)
{
    if (!v || !out || (n_pics && (!n_mbs || !coef_bytes)))
        return fail(MPEGHIP_ERR_INVALID, "stage_begin: NULL argument");
    if (v->stage)
        return fail(MPEGHIP_ERR_INVALID, "stage_begin: the previous stage is still open");
    HIP_TRY(hipSetDevice(v->ctx->device));
    std::unique_ptr<mpeghip_stage> s(new mpeghip_stage);
    s->v = v;
    s->n_pics = n_pics;
    s->mb_first.resize(n_pics);
    s->mb_count.assign(n_mbs, n_mbs + n_pics);
    s->chunk_first.resize(n_pics);
    s->units.resize(n_pics);
    s->alg.assign(n_pics, 0);
    s->blocks.assign(n_pics, 0);
    s->dense.assign(n_pics, 0);
    s->use.assign(n_pics, PicUse());
    s->done.reset(new std::atomic<uint8_t>[n_pics ? n_pics : 1]);
    for (uint32_t i = 0; i < n_pics; i++)
        s->done[i].store(0);
    uint64_t mbs = 0, chunks = 0, words = 0; // words: worst case (every coefficient of every unit non-zero)
    for (uint32_t i = 0; i < n_pics; i++) {
        if (coef_bytes[i] % 4) // (a picture in the unit form: a multiple of 128, checked by its put)
            return fail(MPEGHIP_ERR_INVALID, "stage_begin: picture %u: %zu coefficient bytes is not a multiple of 4", i, coef_bytes[i]);
        s->mb_first[i] = (uint32_t)mbs;
        s->chunk_first[i] = (uint32_t)chunks;
        s->units[i] = coef_bytes[i];
        mbs += n_mbs[i];
        chunks += rc_max_chunks(n_mbs[i]);
        words += words_room(coef_bytes[i], n_mbs[i]);
        if (mbs > 0xffffffffull || words > 0xffffffffull - kRcWordsPad)
            return fail(MPEGHIP_ERR_INVALID, "batch too large for 32-bit indices");
    }
    s->n_mbs = (uint32_t)mbs;
    s->n_chunks = (uint32_t)chunks;
    s->words_cap = words;
    mpeghip_video::Staging *sg = &v->staging[v->next_staging];
    int rc = retire(sg); // two submits ago: normally long finished (a device-packed commit's verdict: here)
    if (rc != MPEGHIP_OK)
        return rc;
    const BlobLayout l = blob_layout(n_pics, chunks);
    s->c_at = l.c_at;
    s->w_at = l.w_at;
    rc = grow_pinned(sg, l.w_at + (size_t)(words + kRcWordsPad) * 4 + 64);
    if (rc != MPEGHIP_OK)
        return rc;
    if (!sg->done)
        HIP_TRY(hipEventCreateWithFlags(&sg->done, hipEventDisableTiming));
    s->sg = sg;
    v->stage = s.get();
    *out = s.release();
    return MPEGHIP_OK;
}

int mpeghip_video_stage_begin(mpeghip_video *v, uint32_t n_pics, const uint32_t *n_mbs, const size_t *coef_bytes,
                              mpeghip_stage **out)
{
    return no_throw([&] { return mpeghip_video_stage_begin_impl(v, n_pics, n_mbs, coef_bytes, out); });
}

static int mpeghip_video_stage_begin_sparse_impl(mpeghip_video *v, uint32_t n_pics, const uint32_t *n_mbs, const size_t *n_words,
                                     mpeghip_stage **out)
{
    if (!n_words && n_pics)
        return fail(MPEGHIP_ERR_INVALID, "stage_begin: NULL argument");
    std::vector<size_t> bytes(n_pics);
    for (uint32_t i = 0; i < n_pics; i++) {
        if (n_words[i] > 0x3fffffffu)
            return fail(MPEGHIP_ERR_INVALID, "batch too large for 32-bit indices");
        bytes[i] = n_words[i] * 4;
    }
    return mpeghip_video_stage_begin(v, n_pics, n_mbs, bytes.data(), out);
}

int mpeghip_video_stage_begin_sparse(mpeghip_video *v, uint32_t n_pics, const uint32_t *n_mbs, const size_t *n_words,
                                     mpeghip_stage **out)
{
    return no_throw([&] { return mpeghip_video_stage_begin_sparse_impl(v, n_pics, n_mbs, n_words, out); });
}

static int stage_put_device(mpeghip_stage *s, uint32_t i, const mpeghip_pic_desc *pic, const mpeghip_mb_desc *mbs, const void *words,
                            bool copy);
// Thread-safe for distinct i: touches only picture i's part of the staging buffer and of the stage's arrays.
static int mpeghip_video_stage_put_impl(mpeghip_stage *s, uint32_t i, const mpeghip_pic_desc *pic, const mpeghip_mb_desc *mbs,
                            const void *coefs)
{
    if (!s || !pic)
        return fail(MPEGHIP_ERR_INVALID, "stage_put: NULL argument");
    const bool sparse = (pic->flags & MPEGHIP_PIC_SPARSE) != 0;
    if (s->device)
        return stage_put_device(s, i, pic, mbs, coefs, true);
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
        if (!sparse && s->units[i] % MPEGHIP_COEF_UNIT) {
            rc = fail(MPEGHIP_ERR_INVALID, "stage_put: picture %u: coef_bytes %llu is not a multiple of 128", i, (unsigned long long)s->units[i]);
            break;
        }
        // one put per picture: the staging buffer has room for each picture once (a second put — a retry after an error, the
        // same i from two threads — would write past it)
        uint8_t fresh = 0;
        if (!s->done[i].compare_exchange_strong(fresh, 1)) {
            rc = fail(MPEGHIP_ERR_INVALID, "stage_put: picture %u was already put", i);
            break;
        }
        if ((rc = validate_pic(v->info, *pic, i)) != MPEGHIP_OK)
            break;
        uint8_t *h = s->sg->h;
        mpeghip_pic_desc pd = *pic;
        pd.mb_first = first;
        pd.mb_count = n;
        reinterpret_cast<mpeghip_pic_desc *>(h)[i] = pd;
        uint64_t alg = 0;
        static thread_local std::vector<uint64_t> seen;
        uint64_t named_units = 0;
        // (sparse: the coefficient extents are in the words themselves: the packer checks them on its way)
        const uint64_t unit_room = sparse ? ~0ull >> 2 : s->units[i] / MPEGHIP_COEF_UNIT;
        if ((rc = validate_picture(v->info, pd, i, mbs, n, 0, unit_room, &alg, &s->use[i], seen, &named_units)) != MPEGHIP_OK)
            break;

        // the picture in the device format: its chunks go where they belong; its words are packed in this thread's
        // scratch memory first, because the room they need is only known afterwards
        static thread_local std::vector<uint32_t> scratch;
        const size_t worst = words_room(s->units[i], n) + 64;
        if (scratch.size() < worst)
            scratch.resize(worst + worst / 4 + 1024);
        uint32_t *chunks = reinterpret_cast<uint32_t *>(h + s->c_at) + (size_t)s->chunk_first[i] * kRcChunkDwords;
        const RcPacked got = sparse ? rc_pack_picture<true, true>(record_geometry(v), pd, mbs, n, static_cast<const uint8_t *>(coefs), 0,
                                                                  chunks, scratch.data(), s->units[i] / 4, worst)
                                    : rc_pack_picture(record_geometry(v), pd, mbs, n, static_cast<const uint8_t *>(coefs), 0, chunks,
                                                      scratch.data());
        if (got.bad) {
            rc = fail(MPEGHIP_ERR_INVALID, "%s", sparse_error_text(i, got.bad - 1).c_str());
            break;
        }
        const uint64_t at = s->words_used.fetch_add(got.words);
        if (at + got.words > s->words_cap) { // (cannot happen: every picture is put once and stays within its worst case)
            rc = fail(MPEGHIP_ERR_INVALID, "stage_put: picture %u: the staging buffer is full", i);
            break;
        }
        memcpy(h + s->w_at + at * 4, scratch.data(), (size_t)got.words * 4);
        rc_rebase(chunks, got.chunks, (uint32_t)at);
        s->alg[i] = alg;
        s->blocks[i] = got.blocks;
        s->dense[i] = got.dense_blocks;
        s->done[i].store(2);
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

int mpeghip_video_stage_put(mpeghip_stage *s, uint32_t i, const mpeghip_pic_desc *pic, const mpeghip_mb_desc *mbs,
                            const void *coefs)
{
    return no_throw([&] { return mpeghip_video_stage_put_impl(s, i, pic, mbs, coefs); });
}

int mpeghip_video_stage_put_sparse(mpeghip_stage *s, uint32_t i, const mpeghip_pic_desc *pic, const mpeghip_mb_desc *mbs,
                                   const uint32_t *words)
{
    if (!pic)
        return fail(MPEGHIP_ERR_INVALID, "stage_put: NULL argument");
    mpeghip_pic_desc p = *pic;
    p.flags |= MPEGHIP_PIC_SPARSE;
    return mpeghip_video_stage_put(s, i, &p, mbs, words);
}

// One picture per call in the sparse form: the single-stream decoder's flush (mpeg::Video).
int mpeghip_video_submit_sparse(mpeghip_video *v, const mpeghip_pic_desc *pic, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                                const uint32_t *words, size_t n_words)
{
    if (!v || !pic)
        return fail(MPEGHIP_ERR_INVALID, "NULL argument");
    if (n_words > 0x3fffffffu)
        return fail(MPEGHIP_ERR_INVALID, "batch too large for 32-bit indices");
    mpeghip_pic_desc p = *pic;
    p.flags |= MPEGHIP_PIC_SPARSE;
    p.mb_first = 0;
    p.mb_count = n_mbs;
    return mpeghip_video_submit(v, &p, 1, mbs, n_mbs, words, n_words * 4);
}

// A staged commit's H2D copy: on the context's copy stream, in pieces (one copy of a gigabyte ran at a quarter of the rate of the
// same bytes in 32-128 MB pieces); the compute stream waits for it.  The destination's previous use has been retired by the
// stage's begin.  What the caller queued on the compute stream before (set_quant, write_planes ...) is not waited for by the
// copy — it only fills a buffer nothing else reads — but by the kernels behind it, as before.
static int send_staging(mpeghip_video *v, mpeghip_video::Staging *sg, uint8_t *d_dst, size_t total)
{
    hipStream_t cs = v->ctx->copy_stream;
    if (!sg->copied)
        HIP_TRY(hipEventCreateWithFlags(&sg->copied, hipEventDisableTiming));
    const size_t piece = (size_t)64 << 20;
    for (size_t at = 0; at < total; at += piece)
        HIP_TRY(hipMemcpyAsync(d_dst + at, sg->h + at, total - at < piece ? total - at : piece, hipMemcpyHostToDevice, cs));
    HIP_TRY(hipEventRecord(sg->copied, cs));
    HIP_TRY(hipStreamWaitEvent(v->ctx->stream, sg->copied, 0));
    return MPEGHIP_OK;
}

// ---- the device-packed stage (include/mpeghip.h: mpeghip_video_stage_begin_device): the host copies, the device packs
static size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static int mpeghip_video_stage_begin_device_impl(mpeghip_video *v, uint32_t n_pics, const uint32_t *n_mbs, const size_t *n_words,
                                                 mpeghip_stage **out)
{
    if (!v || !out || (n_pics && (!n_mbs || !n_words)))
        return fail(MPEGHIP_ERR_INVALID, "stage_begin: NULL argument");
    if (v->stage)
        return fail(MPEGHIP_ERR_INVALID, "stage_begin: the previous stage is still open");
    HIP_TRY(hipSetDevice(v->ctx->device));
    std::unique_ptr<mpeghip_stage> s(new mpeghip_stage);
    s->v = v;
    s->device = true;
    s->n_pics = n_pics;
    s->mb_first.resize(n_pics);
    s->mb_count.assign(n_mbs, n_mbs + n_pics);
    s->chunk_first.resize(n_pics);
    s->word_first.resize(n_pics);
    s->units.resize(n_pics);
    s->use.assign(n_pics, PicUse());
    s->done.reset(new std::atomic<uint8_t>[n_pics ? n_pics : 1]);
    for (uint32_t i = 0; i < n_pics; i++)
        s->done[i].store(0);
    uint64_t mbs = 0, chunks = 0, words = 0;
    for (uint32_t i = 0; i < n_pics; i++) {
        if (n_words[i] > 0x3fffffffu)
            return fail(MPEGHIP_ERR_INVALID, "batch too large for 32-bit indices");
        s->mb_first[i] = (uint32_t)mbs;
        s->chunk_first[i] = (uint32_t)chunks;
        s->word_first[i] = (uint32_t)words;
        s->units[i] = (uint64_t)n_words[i] * 4;
        mbs += n_mbs[i];
        chunks += rc_max_chunks(n_mbs[i]);
        words += round_up(n_words[i], 16); // (a picture's words begin on a cache-line quarter: 64 bytes)
        if (mbs > 0xffffffffull || words > 0xffffffffull - kRcWordsPad)
            return fail(MPEGHIP_ERR_INVALID, "batch too large for 32-bit indices");
    }
    s->n_mbs = (uint32_t)mbs;
    s->n_chunks = (uint32_t)chunks;
    s->words_total = words;
    mpeghip_video::Staging *sg = &v->staging[v->next_staging];
    int rc = retire(sg);
    if (rc != MPEGHIP_OK)
        return rc;
    // pinned: pictures | PkPic | dependency list | macroblocks | words, every region on a 64-byte boundary
    s->a_at = round_up(sizeof(mpeghip_pic_desc) * (size_t)n_pics, 64);
    s->d_at = round_up(s->a_at + sizeof(PkPic) * (size_t)n_pics, 64);
    s->m_at = round_up(s->d_at + sizeof(PkDep) * (size_t)n_pics, 64);
    s->in_at = round_up(s->m_at + sizeof(mpeghip_mb_desc) * (size_t)mbs, 64);
    if ((rc = grow_pinned(sg, s->in_at + (size_t)words * 4 + 64)) != MPEGHIP_OK)
        return rc;
    if (!sg->done)
        HIP_TRY(hipEventCreateWithFlags(&sg->done, hipEventDisableTiming));
    PkPic *aux = reinterpret_cast<PkPic *>(sg->h + s->a_at);
    for (uint32_t i = 0; i < n_pics; i++)
        aux[i] = PkPic{s->word_first[i], (uint32_t)n_words[i], s->chunk_first[i], 0u};
    s->sg = sg;
    v->stage = s.get();
    *out = s.release();
    return MPEGHIP_OK;
}

// A put's copy into the pinned staging buffer, with NON-TEMPORAL stores.  A plain memcpy leaves the lines dirty in the
// core's caches: the H2D copy that follows then has to get them out of there line by line (the DMA engine's reads are
// snooped), and it reads for ownership what it is about to overwrite.  Measured with 8 putting threads: 43 GB/s over PCIe
// with memcpy against 63 GB/s when nothing had touched the buffer since it was written (profiles/round4_b_*, round4_d_*: the staged hand-over sweeps).  Streaming stores go
// to memory past the caches; the fence at the end orders them before the commit's copy is queued.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) static void stream_copy_avx2(uint8_t *dst, const uint8_t *src, size_t n)
{
    while (n && (reinterpret_cast<uintptr_t>(dst) & 31)) { // (staging regions start on 32-byte boundaries at least: rarely)
        *dst++ = *src++;
        n--;
    }
    for (; n >= 128; n -= 128, dst += 128, src += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + 96), d);
    }
    for (; n >= 32; n -= 32, dst += 32, src += 32)
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst), _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src)));
    if (n)
        memcpy(dst, src, n);
    _mm_sfence();
}
#endif
static void staging_copy(void *dst, const void *src, size_t n)
{
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 4096) {
        stream_copy_avx2(static_cast<uint8_t *>(dst), static_cast<const uint8_t *>(src), n);
        return;
    }
#endif
    memcpy(dst, src, n);
}

// picture i of a device-packed stage: its descriptor goes into the staging buffer; its arrays are copied there (copy) or
// are there already (the caller wrote them through mpeghip_video_stage_map)
static int stage_put_device(mpeghip_stage *s, uint32_t i, const mpeghip_pic_desc *pic, const mpeghip_mb_desc *mbs, const void *words,
                            bool copy)
{
    int rc = MPEGHIP_OK;
    do {
        if (i >= s->n_pics) {
            rc = fail(MPEGHIP_ERR_INVALID, "stage_put: picture %u of %u", i, s->n_pics);
            break;
        }
        if (!(pic->flags & MPEGHIP_PIC_SPARSE)) {
            rc = fail(MPEGHIP_ERR_INVALID, "stage_put: picture %u: a device-packed stage takes pictures in the sparse form only "
                      "(mpeghip_video_stage_put_sparse)", i);
            break;
        }
        const uint32_t n = s->mb_count[i];
        if (copy && ((n && !mbs) || (s->units[i] && !words))) {
            rc = fail(MPEGHIP_ERR_INVALID, "stage_put: picture %u: NULL array", i);
            break;
        }
        uint8_t fresh = 0;
        if (!s->done[i].compare_exchange_strong(fresh, 1)) {
            rc = fail(MPEGHIP_ERR_INVALID, "stage_put: picture %u was already put", i);
            break;
        }
        if ((rc = validate_pic(s->v->info, *pic, i)) != MPEGHIP_OK)
            break;
        uint8_t *h = s->sg->h;
        mpeghip_pic_desc pd = *pic;
        pd.mb_first = s->mb_first[i];
        pd.mb_count = n;
        reinterpret_cast<mpeghip_pic_desc *>(h)[i] = pd;
        if (copy) {
            staging_copy(h + s->m_at + (size_t)s->mb_first[i] * sizeof(mpeghip_mb_desc), mbs, (size_t)n * sizeof(mpeghip_mb_desc));
            staging_copy(h + s->in_at + (size_t)s->word_first[i] * 4, words, (size_t)s->units[i]);
        }
        s->done[i].store(2);
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

// the commit of a device-packed stage: one copy of the staged arrays, pack_kernel, pack_gate_kernel, the reconstruction
static int stage_commit_device(mpeghip_stage *s)
{
    mpeghip_video *v = s->v;
    mpeghip_video::Staging *sg = s->sg;
    mpeghip_batch *b = &sg->batch;
    const mpeghip_video_info &in = v->info;
    const mpeghip_pic_desc *pics = reinterpret_cast<const mpeghip_pic_desc *>(sg->h);
    // Pictures of one stream inside one commit run concurrently (check_dependencies): two that write the same slot are
    // refused here; one that READS a slot another one writes only if a macroblock really predicts from it — which the device
    // learns while packing: the candidates go along as a list
    PkDep *deps = reinterpret_cast<PkDep *>(sg->h + s->d_at);
    uint32_t n_deps = 0;
    if (s->n_pics > 1) {
        std::vector<uint32_t> order(s->n_pics);
        for (uint32_t p = 0; p < s->n_pics; p++)
            order[p] = p;
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            return pics[x].stream != pics[y].stream ? pics[x].stream < pics[y].stream : x < y;
        });
        for (uint32_t i = 0; i < s->n_pics;) {
            uint32_t j = i + 1;
            while (j < s->n_pics && pics[order[j]].stream == pics[order[i]].stream)
                j++;
            for (uint32_t y = i; y < j && j - i > 1; y++) {
                const mpeghip_pic_desc &r = pics[order[y]];
                uint32_t mask = 0;
                for (uint32_t x = i; x < j; x++) {
                    if (x == y)
                        continue;
                    const mpeghip_pic_desc &w = pics[order[x]];
                    if (x < y && w.cur == r.cur)
                        return fail(MPEGHIP_ERR_INVALID, "pictures %u and %u of stream %u depend on each other (slot %u): they need "
                                    "separate submits", order[x], order[y], w.stream, w.cur);
                    mask |= (r.fwd == w.cur ? 1u : 0u) | (r.bwd == w.cur ? 2u : 0u);
                }
                if (mask)
                    deps[n_deps++] = PkDep{order[y], mask};
            }
            i = j;
        }
    }
    int rc;
    b->any_rgba = wants_rgba(pics, s->n_pics);
    if (b->any_rgba && (rc = ensure_rgba(v)) != MPEGHIP_OK)
        return rc;
    const size_t raw_total = s->in_at + (size_t)s->words_total * 4;
    const size_t c_bytes = round_up((size_t)s->n_chunks * kRcChunkDwords * 4, 64);
    const uint32_t seen_stride = (in.mb_w * in.mb_h + 31) / 32;
    if ((rc = grow((void **)&sg->d_raw, &sg->cap_raw, raw_total + 64)) != 0 ||
        (rc = grow((void **)&b->d_blob, &b->cap_blob, c_bytes + ((size_t)s->words_total + kRcWordsPad) * 4)) != 0 ||
        (rc = grow((void **)&sg->d_seen, &sg->cap_seen, (size_t)s->n_pics * seen_stride * 4)) != 0)
        return rc;
    // per picture its report word, then the gate's scratch: [n] first report, [n + 1] refused pictures, [n + 2] workgroups done
    if ((rc = grow((void **)&sg->d_err, &sg->cap_err, ((size_t)s->n_pics + 3) * 8)) != 0)
        return rc;
    if (!sg->h_verdict)
        HIP_TRY(hipHostMalloc((void **)&sg->h_verdict, 16 + kPkVerdictList * 4, hipHostMallocDefault));
    if (!sg->gated)
        HIP_TRY(hipEventCreateWithFlags(&sg->gated, hipEventDisableTiming));
    sg->h_verdict[0] = kPkNoError;
    sg->h_verdict[1] = 0;
    sg->owner = v;
    hipStream_t st = v->ctx->stream;
    // (the packer's scratch is reset while the copy runs; the kernels wait for the copy)
    HIP_TRY(hipMemsetAsync(sg->d_seen, 0, (size_t)s->n_pics * seen_stride * 4, st));
    HIP_TRY(hipMemsetAsync(sg->d_err, 0xff, ((size_t)s->n_pics + 1) * 8, st));
    HIP_TRY(hipMemsetAsync(sg->d_err + s->n_pics + 1, 0, 16, st));
    if ((rc = send_staging(v, sg, sg->d_raw, raw_total)) != MPEGHIP_OK) {
        (void)hipStreamSynchronize(v->ctx->copy_stream); // (what was queued of the copy has read the staging buffer)
        return rc;
    }
    // From here on the H2D copy is reading sg->h on the copy stream.  Whatever fails below, the call must not return with that
    // copy still in flight and the slot marked idle: the next stage_begin would refill the buffer under it (round-4 advisor).
    rc = [&]() -> int {
    PackArgs a;
    a.pics = reinterpret_cast<const mpeghip_pic_desc *>(sg->d_raw);
    a.aux = reinterpret_cast<PkPic *>(sg->d_raw + s->a_at);
    a.mbs = reinterpret_cast<const mpeghip_mb_desc *>(sg->d_raw + s->m_at);
    a.words_in = reinterpret_cast<const uint32_t *>(sg->d_raw + s->in_at);
    a.chunks = reinterpret_cast<uint32_t *>(b->d_blob);
    a.words_out = reinterpret_cast<uint32_t *>(b->d_blob + c_bytes);
    a.seen = sg->d_seen;
    a.err = sg->d_err;
    a.n_pics = s->n_pics;
    uint32_t most = 1;
    for (uint32_t i = 0; i < s->n_pics; i++)
        most = s->mb_count[i] > most ? s->mb_count[i] : most;
    a.groups_per_pic = (most + 63) / 64;
    a.seen_stride = seen_stride;
    a.mb_w = in.mb_w;
    a.mb_h = in.mb_h;
    a.luma_w = in.luma_w;
    a.chroma_w = in.chroma_w;
    a.luma_bytes = (uint32_t)in.luma_bytes;
    a.chroma_bytes = (uint32_t)in.chroma_bytes;
    a.frame_bytes = in.frame_bytes;
    a.frame_stride = in.frame_stride;
    a.rgba_stride = rgba_stride_of(v);
    if ((uint64_t)s->n_pics * a.groups_per_pic > 0x7fffffffull)
        return fail(MPEGHIP_ERR_INVALID, "batch too large for one launch");
    {   // the window: twice what a wave of the picture with the most words per macroblock names on average, 512 .. kPkWinDwords
        uint64_t most_per_wave = 0;
        for (uint32_t i = 0; i < s->n_pics; i++)
            if (s->mb_count[i]) {
                const uint64_t per_wave = s->units[i] / 4 * 64 / s->mb_count[i];
                most_per_wave = per_wave > most_per_wave ? per_wave : most_per_wave;
            }
        uint64_t win = (2 * most_per_wave + 255) / 256 * 256;
        win = win < 512 ? 512 : (win > kPkWinDwords ? kPkWinDwords : win);
        a.win_dwords = (uint32_t)win;
    }
    hipLaunchKernelGGL(pack_kernel, dim3(s->n_pics * a.groups_per_pic), dim3(64), (64 * kPkXchDwords + 2 * a.win_dwords) * 4, st, a);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(pack_gate_kernel, dim3(s->n_pics), dim3(256), 0, st, sg->d_err, a.aux, a.pics,
                       reinterpret_cast<const PkDep *>(sg->d_raw + s->d_at), n_deps, a.chunks, s->n_pics, sg->d_err + s->n_pics, sg->h_verdict);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sg->gated, st));
    fill_notes(v, b, pics, s->n_pics);
    b->d_pics = reinterpret_cast<mpeghip_pic_desc *>(sg->d_raw);
    b->d_chunks = a.chunks;
    b->d_words = a.words_out;
    // which instance of the reconstruction kernel: the host has not looked at the blocks; their words tell (a dense unit
    // arrives as more than 32 pairs, a typical macroblock as about 16 dwords: kDenseWordsPerMb lies between the two)
    uint64_t words_in = 0;
    for (uint32_t i = 0; i < s->n_pics; i++)
        words_in += s->units[i] / 4;
    b->coded_blocks = 1;
    b->dense_blocks = words_in > kDenseWordsPerMb * (uint64_t)s->n_mbs ? 1 : 0;
    b->alg_bytes = 0; // (not totalled: nobody has read the macroblocks on the host)
    b->device_bytes = raw_total;
    b->replicas = 1;
    b->n_pics = s->n_pics;
    b->n_mbs = s->n_mbs;
    b->n_chunks = s->n_chunks;
    v->next_staging ^= 1;
    int rc2;
    if ((rc2 = launch_batch(v, b)) != MPEGHIP_OK)
        return rc2;
    HIP_TRY(hipEventRecord(sg->done, st));
    sg->in_flight = true;
    sg->packed_on_device = true;
    sg->pk_mb_first = s->mb_first;
    sg->pk_stream.resize(s->n_pics);
    for (uint32_t i = 0; i < s->n_pics; i++)
        sg->pk_stream[i] = pics[i].stream;
    return MPEGHIP_OK;
    }();
    if (rc != MPEGHIP_OK) { // keep the error text: the synchronisation calls below do not touch it
        (void)hipStreamSynchronize(v->ctx->copy_stream);
        (void)hipStreamSynchronize(st);
    }
    return rc;
}

static int mpeghip_video_stage_commit_impl(mpeghip_stage *sp)
{
    if (!sp)
        return fail(MPEGHIP_ERR_INVALID, "stage_commit: NULL stage");
    std::unique_ptr<mpeghip_stage> s(sp); // the stage ends here, whatever happens
    mpeghip_video *v = s->v;
    v->stage = nullptr;
    if (s->error.load() != MPEGHIP_OK)
        return fail(s->error.load(), "%s", s->error_text.c_str());
    for (uint32_t i = 0; i < s->n_pics; i++)
        if (s->done[i].load() != 2)
            return fail(MPEGHIP_ERR_INVALID, "stage_commit: picture %u was never put", i);
    if (s->n_mbs == 0)
        return MPEGHIP_OK;
    HIP_TRY(hipSetDevice(v->ctx->device));
    if (s->device)
        return stage_commit_device(s.get());
    mpeghip_video::Staging *sg = s->sg;
    mpeghip_batch *b = &sg->batch;
    const mpeghip_pic_desc *pics = reinterpret_cast<const mpeghip_pic_desc *>(sg->h);
    int rc = check_dependencies(pics, s->use.data(), s->n_pics);
    if (rc != MPEGHIP_OK)
        return rc;
    b->any_rgba = wants_rgba(pics, s->n_pics);
    if (b->any_rgba && (rc = ensure_rgba(v)) != MPEGHIP_OK)
        return rc;
    // the device image of the staging buffer: pictures | chunks | words, sent as ONE range
    // (a copy costs the better part of a millisecond of stream time whatever its size)
    const size_t total = s->w_at + (size_t)s->words_used.load() * 4;
    if ((rc = grow((void **)&b->d_blob, &b->cap_blob, total + kRcWordsPad * 4)) != 0)
        return rc;
    b->d_pics = reinterpret_cast<mpeghip_pic_desc *>(b->d_blob);
    b->d_chunks = reinterpret_cast<uint32_t *>(b->d_blob + s->c_at);
    b->d_words = reinterpret_cast<uint32_t *>(b->d_blob + s->w_at);
    hipStream_t st = v->ctx->stream;
    if ((rc = send_staging(v, sg, b->d_blob, total)) != MPEGHIP_OK)
        return rc;
    fill_notes(v, b, pics, s->n_pics);
    b->alg_bytes = b->coded_blocks = b->dense_blocks = 0;
    for (uint32_t p = 0; p < s->n_pics; p++) {
        b->alg_bytes += s->alg[p];
        b->coded_blocks += s->blocks[p];
        b->dense_blocks += s->dense[p];
    }
    b->replicas = 1;
    b->n_pics = s->n_pics;
    b->n_mbs = s->n_mbs;
    b->n_chunks = s->n_chunks;
    v->next_staging ^= 1;
    rc = launch_batch(v, b);
    if (rc != MPEGHIP_OK)
        return rc;
    HIP_TRY(hipEventRecord(sg->done, st));
    sg->in_flight = true;
    return MPEGHIP_OK;
}

int mpeghip_video_stage_commit(mpeghip_stage *sp)
{
    return no_throw([&] { return mpeghip_video_stage_commit_impl(sp); });
}

int mpeghip_video_stage_begin_device(mpeghip_video *v, uint32_t n_pics, const uint32_t *n_mbs, const size_t *n_words, mpeghip_stage **out)
{
    return no_throw([&] { return mpeghip_video_stage_begin_device_impl(v, n_pics, n_mbs, n_words, out); });
}

int mpeghip_video_stage_map(mpeghip_stage *s, uint32_t i, mpeghip_mb_desc **mbs, uint32_t **words)
{
    if (!s || !mbs || !words)
        return fail(MPEGHIP_ERR_INVALID, "stage_map: NULL argument");
    if (!s->device)
        return fail(MPEGHIP_ERR_INVALID, "stage_map: not a device-packed stage (mpeghip_video_stage_begin_device)");
    if (i >= s->n_pics)
        return fail(MPEGHIP_ERR_INVALID, "stage_map: picture %u of %u", i, s->n_pics);
    *mbs = reinterpret_cast<mpeghip_mb_desc *>(s->sg->h + s->m_at) + s->mb_first[i];
    *words = reinterpret_cast<uint32_t *>(s->sg->h + s->in_at) + s->word_first[i];
    return MPEGHIP_OK;
}

int mpeghip_video_stage_put_mapped(mpeghip_stage *s, uint32_t i, const mpeghip_pic_desc *pic)
{
    if (!s || !pic)
        return fail(MPEGHIP_ERR_INVALID, "stage_put: NULL argument");
    if (!s->device)
        return fail(MPEGHIP_ERR_INVALID, "stage_put_mapped: not a device-packed stage (mpeghip_video_stage_begin_device)");
    mpeghip_pic_desc p = *pic;
    p.flags |= MPEGHIP_PIC_SPARSE;
    return no_throw([&] { return stage_put_device(s, i, &p, nullptr, nullptr, false); });
}

int mpeghip_video_sync(mpeghip_video *v)
{
    if (!v)
        return fail(MPEGHIP_ERR_INVALID, "video is NULL");
    HIP_TRY(hipSetDevice(v->ctx->device));
    HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    for (auto &sg : v->staging)
        sg.in_flight = false;
    return reap_all(v);
}

// Wait until the device has VALIDATED every device-packed commit queued so far — pack_kernel + pack_gate_kernel: a fraction of the
// commit's time on the device — not until it has reconstructed them; the deferred refusal, if any, like mpeghip_video_sync.
int mpeghip_video_verdict(mpeghip_video *v)
{
    if (!v)
        return fail(MPEGHIP_ERR_INVALID, "video is NULL");
    HIP_TRY(hipSetDevice(v->ctx->device));
    for (int i = 0; i < 2; i++) { // older first
        mpeghip_video::Staging *sg = &v->staging[v->next_staging ^ i];
        if (!sg->packed_on_device)
            continue;
        if (sg->in_flight)
            HIP_TRY(hipEventSynchronize(sg->gated));
        const int rc = reap_verdict(sg);
        if (rc != MPEGHIP_OK)
            return rc;
    }
    return MPEGHIP_OK;
}

uint64_t mpeghip_video_refused(const mpeghip_video *v, uint32_t *pics, uint32_t *streams, uint32_t cap)
{
    if (!v)
        return 0;
    for (size_t i = 0; i < v->refused.size() && i < cap; i++) {
        if (pics)
            pics[i] = v->refused[i].first;
        if (streams)
            streams[i] = v->refused[i].second;
    }
    return v->refused_total;
}

static int mpeghip_video_batch_upload_replicated_impl(mpeghip_video *v, const mpeghip_pic_desc *pics, uint32_t n_pics,
                                          const mpeghip_mb_desc *mbs, uint32_t n_mbs, const void *coefs,
                                          size_t coef_bytes, uint32_t n_streams, mpeghip_batch **out)
{
    if (!v || !out)
        return fail(MPEGHIP_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_streams == 0 || n_streams > v->info.n_streams)
        return fail(MPEGHIP_ERR_INVALID, "n_streams %u out of range", n_streams);
    if (n_pics && !pics)
        return fail(MPEGHIP_ERR_INVALID, "pics is NULL");
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

int mpeghip_video_batch_upload_replicated(mpeghip_video *v, const mpeghip_pic_desc *pics, uint32_t n_pics,
                                          const mpeghip_mb_desc *mbs, uint32_t n_mbs, const void *coefs,
                                          size_t coef_bytes, uint32_t n_streams, mpeghip_batch **out)
{
    return no_throw([&] { return mpeghip_video_batch_upload_replicated_impl(v, pics, n_pics, mbs, n_mbs, coefs, coef_bytes, n_streams, out); });
}

static int mpeghip_video_batch_upload_impl(mpeghip_video *v, const mpeghip_pic_desc *pics, uint32_t n_pics,
                               const mpeghip_mb_desc *mbs, uint32_t n_mbs, const void *coefs, size_t coef_bytes,
                               mpeghip_batch **out)
{
    return mpeghip_video_batch_upload_replicated(v, pics, n_pics, mbs, n_mbs, coefs, coef_bytes, 1, out);
}

int mpeghip_video_batch_upload(mpeghip_video *v, const mpeghip_pic_desc *pics, uint32_t n_pics,
                               const mpeghip_mb_desc *mbs, uint32_t n_mbs, const void *coefs, size_t coef_bytes,
                               mpeghip_batch **out)
{
    return no_throw([&] { return mpeghip_video_batch_upload_impl(v, pics, n_pics, mbs, n_mbs, coefs, coef_bytes, out); });
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
uint64_t mpeghip_video_batch_device_bytes(const mpeghip_batch *b) { return b ? b->device_bytes : 0; }

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

static int ensure_linear(mpeghip_video *v)
{
    if (!v->d_linear)
        HIP_TRY(hipMalloc((void **)&v->d_linear, v->info.frame_bytes + 64));
    return MPEGHIP_OK;
}

static void launch_relayout(mpeghip_video *v, uint8_t *slot, uint32_t first, uint32_t n_bytes, int to_linear)
{
    if (n_bytes)
        hipLaunchKernelGGL(relayout_kernel, dim3((n_bytes / 4 + 255) / 256), dim3(256), 0, v->ctx->stream, slot, v->d_linear + first, first,
                           n_bytes, v->info.mb_w, (uint32_t)v->info.luma_bytes, (uint32_t)v->info.chroma_bytes, to_linear);
}

int mpeghip_video_read_planes(mpeghip_video *v, uint32_t stream, uint32_t slot, uint8_t *y, uint8_t *cb, uint8_t *cr)
{
    if (!v || stream >= v->info.n_streams || slot >= MPEGHIP_SLOTS)
        return fail(MPEGHIP_ERR_INVALID, "bad stream/slot");
    HIP_TRY(hipSetDevice(v->ctx->device));
    int rc = ensure_linear(v);
    if (rc != MPEGHIP_OK)
        return rc;
    // the planes are tiled in HBM (video_lane.h): untile into the linear scratch, then ONE copy of Y, Cb, Cr into a
    // pinned bounce buffer behind everything queued on the stream, then plain memcpys (three synchronous copies
    // into pageable memory cost three round trips — most of a small picture's turnaround)
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
    launch_relayout(v, slot_ptr(v, stream, slot), 0, (uint32_t)bytes, 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(v->bounce, v->d_linear, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if ((rc = reap_all(v)) != MPEGHIP_OK) // a device-packed commit that was refused: the caller learns it here at the latest
        return rc;
    if (y)
        memcpy(y, v->bounce, v->info.luma_bytes);
    if (cb)
        memcpy(cb, v->bounce + v->info.luma_bytes, v->info.chroma_bytes);
    if (cr)
        memcpy(cr, v->bounce + v->info.luma_bytes + v->info.chroma_bytes, v->info.chroma_bytes);
    return MPEGHIP_OK;
}

// ---- the asynchronous read-back (ABI 3): what lets a lone decoder parse picture N + 1 while picture N is on the device
int mpeghip_video_read_planes_async(mpeghip_video *v, uint32_t stream, uint32_t slot, uint8_t *dst, uint64_t *ticket)
{
    if (!v || !dst || !ticket || stream >= v->info.n_streams || slot >= MPEGHIP_SLOTS)
        return fail(MPEGHIP_ERR_INVALID, "read_planes_async: bad argument");
    HIP_TRY(hipSetDevice(v->ctx->device));
    hipStream_t st = v->ctx->stream;
    const size_t bytes = v->info.luma_bytes + 2 * v->info.chroma_bytes;
    // `dst` is pinned memory of mpeghip_pinned_alloc: the untiling kernel stores the linear planes straight into it (the device
    // sees pinned host memory) — ONE launch, no copy call; any other memory takes the untile + copy of mpeghip_video_read_planes
    void *d_dst = nullptr;
    if (hipHostGetDevicePointer(&d_dst, dst, 0) == hipSuccess && d_dst) {
        hipLaunchKernelGGL(relayout_kernel, dim3(((uint32_t)bytes / 4 + 255) / 256), dim3(256), 0, st, slot_ptr(v, stream, slot),
                           static_cast<uint8_t *>(d_dst), 0u, (uint32_t)bytes, v->info.mb_w, (uint32_t)v->info.luma_bytes,
                           (uint32_t)v->info.chroma_bytes, 1);
        HIP_TRY(hipGetLastError());
    } else {
        (void)hipGetLastError();
        int rc = ensure_linear(v);
        if (rc != MPEGHIP_OK)
            return rc;
        launch_relayout(v, slot_ptr(v, stream, slot), 0, (uint32_t)bytes, 1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(dst, v->d_linear, bytes, hipMemcpyDeviceToHost, st));
    }
    hipEvent_t &ev = v->read_done[v->reads_issued & 3];
    if (!ev)
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev, st));
    *ticket = v->reads_issued++;
    return MPEGHIP_OK;
}

int mpeghip_video_host_mirror(mpeghip_video *v, int on)
{
    if (!v)
        return fail(MPEGHIP_ERR_INVALID, "host_mirror: no store");
    HIP_TRY(hipSetDevice(v->ctx->device));
    if (!on) {
        if (v->h_mirror) {
            HIP_TRY(hipStreamSynchronize(v->ctx->stream)); // (launches that write it)
            (void)hipHostFree(v->h_mirror);
        }
        v->h_mirror = v->d_mirror = nullptr;
        v->mirror_valid.clear();
        return MPEGHIP_OK;
    }
    if (v->h_mirror)
        return MPEGHIP_OK;
    const uint64_t stride = align_up(v->info.luma_bytes + 2 * v->info.chroma_bytes, 256);
    const uint64_t total = stride * MPEGHIP_SLOTS * v->info.n_streams;
    if (total > (1ull << 30)) // a lone decoder's store (or a few): a thousand streams' frames do not belong in pinned host memory
        return fail(MPEGHIP_ERR_INVALID, "host_mirror: %llu bytes of pinned memory for %u streams (limit 1 GiB)", (unsigned long long)total,
                    v->info.n_streams);
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, total, hipHostMallocDefault) != hipSuccess)
        return fail(MPEGHIP_ERR_OOM, "host_mirror: hipHostMalloc(%llu) failed", (unsigned long long)total);
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) {
        (void)hipGetLastError();
        (void)hipHostFree(h);
        return fail(MPEGHIP_ERR_HIP, "host_mirror: pinned memory is not visible to the device");
    }
    v->h_mirror = static_cast<uint8_t *>(h);
    v->d_mirror = static_cast<uint8_t *>(d);
    v->mirror_stride = stride;
    v->mirror_valid.assign((size_t)v->info.n_streams * MPEGHIP_SLOTS, 0); // (every slot's first request untiles it once)
    return MPEGHIP_OK;
}

int mpeghip_video_mirror_async(mpeghip_video *v, uint32_t stream, uint32_t slot, const uint8_t **planes, uint64_t *ticket)
{
    if (!v || !planes || !ticket || stream >= v->info.n_streams || slot >= MPEGHIP_SLOTS)
        return fail(MPEGHIP_ERR_INVALID, "mirror_async: bad argument");
    if (!v->h_mirror)
        return fail(MPEGHIP_ERR_INVALID, "mirror_async: the store has no host mirror (mpeghip_video_host_mirror)");
    HIP_TRY(hipSetDevice(v->ctx->device));
    hipStream_t st = v->ctx->stream;
    const size_t at = ((size_t)stream * MPEGHIP_SLOTS + slot);
    v->mirror_requests++;
    if (!v->mirror_valid[at]) { // something other than a mirroring launch wrote the slot (or nothing has yet): untile it once
        v->mirror_repairs++;
        const size_t bytes = v->info.luma_bytes + 2 * v->info.chroma_bytes;
        hipLaunchKernelGGL(relayout_kernel, dim3(((uint32_t)bytes / 4 + 255) / 256), dim3(256), 0, st, slot_ptr(v, stream, slot),
                           v->d_mirror + at * v->mirror_stride, 0u, (uint32_t)bytes, v->info.mb_w, (uint32_t)v->info.luma_bytes,
                           (uint32_t)v->info.chroma_bytes, 1);
        HIP_TRY(hipGetLastError());
        v->mirror_valid[at] = 1;
    }
    hipEvent_t &ev = v->read_done[v->reads_issued & 3];
    if (!ev)
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev, st));
    *ticket = v->reads_issued++;
    *planes = v->h_mirror + at * v->mirror_stride;
    return MPEGHIP_OK;
}

void mpeghip_video_mirror_counters(const mpeghip_video *v, uint64_t out[2])
{
    out[0] = v ? v->mirror_requests : 0;
    out[1] = v ? v->mirror_repairs : 0;
}

int mpeghip_video_read_wait(mpeghip_video *v, uint64_t ticket)
{
    if (!v || ticket >= v->reads_issued)
        return fail(MPEGHIP_ERR_INVALID, "read_wait: no such read-back");
    HIP_TRY(hipSetDevice(v->ctx->device));
    HIP_TRY(hipEventSynchronize(v->read_done[ticket & 3])); // (this read-back's event, or a later one's: both say it is done)
    return reap_all(v); // a device-packed commit that was refused: the caller learns it here at the latest
}

int mpeghip_video_write_planes(mpeghip_video *v, uint32_t stream, uint32_t slot, const uint8_t *y, const uint8_t *cb,
                               const uint8_t *cr, const uint8_t *pad)
{
    if (!v || stream >= v->info.n_streams || slot >= MPEGHIP_SLOTS)
        return fail(MPEGHIP_ERR_INVALID, "bad stream/slot");
    v->rgba_sync[(size_t)stream * MPEGHIP_SLOTS + slot] = 0; // the slot's RGBA image is out of date now
    if (v->h_mirror)
        v->mirror_valid[(size_t)stream * MPEGHIP_SLOTS + slot] = 0; // ... and its host mirror
    HIP_TRY(hipSetDevice(v->ctx->device));
    int rc = ensure_linear(v);
    if (rc != MPEGHIP_OK)
        return rc;
    hipStream_t st = v->ctx->stream;
    HIP_TRY(hipStreamSynchronize(st));
    uint8_t *p = slot_ptr(v, stream, slot);
    const uint32_t L = (uint32_t)v->info.luma_bytes, Cb = (uint32_t)v->info.chroma_bytes;
    if (y) {
        HIP_TRY(hipMemcpy(v->d_linear, y, L, hipMemcpyHostToDevice));
        launch_relayout(v, p, 0, L, 0);
    }
    if (cb) {
        HIP_TRY(hipMemcpy(v->d_linear + L, cb, Cb, hipMemcpyHostToDevice));
        launch_relayout(v, p, L, Cb, 0);
    }
    if (cr) {
        HIP_TRY(hipMemcpy(v->d_linear + L + Cb, cr, Cb, hipMemcpyHostToDevice));
        launch_relayout(v, p, L + Cb, Cb, 0);
    }
    HIP_TRY(hipGetLastError());
    if (pad) // (the pad stays linear)
        HIP_TRY(hipMemcpyAsync(p + L + 2 * Cb, pad, (size_t)v->info.luma_w * 16, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MPEGHIP_OK;
}

int mpeghip_video_broadcast_slot(mpeghip_video *v, uint32_t src, uint32_t slot, uint32_t dst0, uint32_t n)
{
    if (!v || src >= v->info.n_streams || slot >= MPEGHIP_SLOTS || (uint64_t)dst0 + n > v->info.n_streams)
        return fail(MPEGHIP_ERR_INVALID, "bad stream/slot");
    HIP_TRY(hipSetDevice(v->ctx->device));
    for (uint32_t s = dst0; s < dst0 + n; s++)
        if (s != src) {
            v->rgba_sync[(size_t)s * MPEGHIP_SLOTS + slot] = 0;
            if (v->h_mirror)
                v->mirror_valid[(size_t)s * MPEGHIP_SLOTS + slot] = 0;
        }
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
    hipLaunchKernelGGL(hash_kernel, dim3((v->info.n_streams + 63) / 64), dim3(64), 0, v->ctx->stream, v->d_frames,
                       v->info.frame_stride, slot, v->info.mb_w, (uint32_t)v->info.luma_bytes, (uint32_t)v->info.chroma_bytes,
                       v->info.n_streams, v->d_hash);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    const int verdict = reap_all(v);
    if (verdict != MPEGHIP_OK)
        return verdict;
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
    // grid.z is limited to 65535
    for (uint32_t s0 = 0; s0 < n; s0 += 32768) {
        const uint32_t ns = n - s0 < 32768 ? n - s0 : 32768;
        dim3 grid((in.mb_w + 7) / 8, in.mb_h, ns);
        hipLaunchKernelGGL(rgba_kernel, grid, dim3(256), 0, v->ctx->stream, v->d_frames, in.frame_stride, v->d_rgba,
                           rgba_stride_of(v), in.mb_w, (uint32_t)in.luma_bytes, (uint32_t)in.chroma_bytes,
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
    if ((rc = reap_all(v)) != MPEGHIP_OK)
        return rc;
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
    (void)hipDeviceGetAttribute(&a->n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
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
    for (auto &e : a->synth_done)
        if (e)
            (void)hipEventDestroy(e);
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
    // Time slices per stream.  5 workgroups stay resident per CU (5 x 30 780 B of LDS fit a CU's 160 000) and all of a launch's
    // workgroups take the same time, so the launch runs in ceil(workgroups / resident) rounds of one slice each; a slice
    // costs its frames plus about a quarter of a frame (rebuilding 15 history slots, state, filling the pipeline).  Take
    // the slice count that minimises rounds x that, among those that leave a slice at least 4 frames: one residency for
    // BASELINE config 4 (256 streams -> 5 slices), 8 full rounds instead of 1.6 -> 2 for 2048 streams.
    uint32_t chunks = 1;
    {
        const uint64_t resident = (uint64_t)(a->n_cu > 0 ? a->n_cu : 1) * 5;
        const uint32_t most = n_frames / 4 < 32 ? n_frames / 4 : 32;
        double best = 0;
        for (uint32_t c = 1; c <= (most ? most : 1); c++) {
            const uint64_t rounds = ((uint64_t)a->n_streams * c + resident - 1) / resident;
            const double cost = (double)rounds * ((double)n_frames / c + 0.25);
            if (c == 1 || cost < best) {
                best = cost;
                chunks = c;
            }
        }
    }
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
    a->can_undo = true;
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

// ---- asynchronous synthesis (ABI 3): a lone Audio decoder parses frame N + 1 while frame N is on the device.  `samples` and `out`
// are pinned memory of mpeghip_pinned_alloc; the kernel reads and writes them in place (the device sees pinned host memory): one
// launch, no copy calls.  Other memory: staged through the handle's device buffers with asynchronous copies.
int mpeghip_audio_synth_async(mpeghip_audio *a, const int32_t *samples, uint32_t n_frames, int format, void *out, uint64_t *ticket)
{
    if (!a || !samples || !out || !ticket || n_frames == 0 || format < 0 || format > MPEGHIP_AUDIO_S16)
        return fail(MPEGHIP_ERR_INVALID, "synth_async: bad argument");
    HIP_TRY(hipSetDevice(a->ctx->device));
    hipStream_t st = a->ctx->stream;
    const size_t n = (size_t)a->n_streams * n_frames * MPEGHIP_AUDIO_FRAME_INTS;
    void *d_in = nullptr, *d_out = nullptr;
    int rc;
    if (hipHostGetDevicePointer(&d_in, const_cast<int32_t *>(samples), 0) == hipSuccess && d_in &&
        hipHostGetDevicePointer(&d_out, out, 0) == hipSuccess && d_out) {
        rc = audio_launch(a, static_cast<const int32_t *>(d_in), n_frames, format, d_out, nullptr);
        if (rc != MPEGHIP_OK)
            return rc;
    } else {
        (void)hipGetLastError();
        if (a->cap_samples < n * sizeof(int32_t) || a->cap_out < n * audio_elem_size(format)) { // (grows: waits for what is queued)
            int32_t *ds;
            void *dout;
            if ((rc = mpeghip_audio_device_buffers(a, n_frames, format, &ds, &dout)) != MPEGHIP_OK)
                return rc;
        }
        HIP_TRY(hipMemcpyAsync(a->d_samples, samples, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
        rc = audio_launch(a, a->d_samples, n_frames, format, a->d_out, nullptr);
        if (rc != MPEGHIP_OK)
            return rc;
        HIP_TRY(hipMemcpyAsync(out, a->d_out, n * audio_elem_size(format), hipMemcpyDeviceToHost, st));
    }
    hipEvent_t &ev = a->synth_done[a->synths_issued & 3];
    if (!ev)
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev, st));
    *ticket = a->synths_issued++;
    return MPEGHIP_OK;
}

int mpeghip_audio_synth_wait(mpeghip_audio *a, uint64_t ticket)
{
    if (!a || ticket >= a->synths_issued)
        return fail(MPEGHIP_ERR_INVALID, "synth_wait: no such launch");
    HIP_TRY(hipSetDevice(a->ctx->device));
    HIP_TRY(hipEventSynchronize(a->synth_done[ticket & 3]));
    return MPEGHIP_OK;
}

// Forget the LAST launch: the synthesis state (V ring, vPos of every stream) is again what it was before it — a launch writes the
// new state into the alternate buffers, so the old one is still there.  For a decoder that synthesised one frame ahead and is
// told to rewind: the reference (audio.go:149-154) keeps the ring as the frames it RETURNED left it.  One level only.
int mpeghip_audio_undo_last(mpeghip_audio *a)
{
    if (!a)
        return fail(MPEGHIP_ERR_INVALID, "NULL argument");
    if (!a->can_undo)
        return fail(MPEGHIP_ERR_INVALID, "undo_last: no launch to undo (one level only)");
    HIP_TRY(hipSetDevice(a->ctx->device));
    HIP_TRY(hipStreamSynchronize(a->ctx->stream));
    float *r = a->d_ring;
    a->d_ring = a->d_ring_alt;
    a->d_ring_alt = r;
    int32_t *vp = a->d_vpos;
    a->d_vpos = a->d_vpos_alt;
    a->d_vpos_alt = vp;
    a->can_undo = false;
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
    a->can_undo = false; // (the state is the caller's from here on)
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
    HIP_TRY(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_phase_dump), bytes));
    return MPEGHIP_OK;
}
#endif

} // extern "C"
