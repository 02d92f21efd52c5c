// video_pack_lane.h — the DEVICE-side packer: what validate_mb / rc_pack_picture (mpeghip.hip, video_recon_lane.h) do on the
// host for a submit, done on the GPU for a DEVICE-PACKED STAGE (mpeghip_video_stage_begin_device, include/mpeghip.h).
//
// Why: the host packer costs 0.27 ms of one core per typical 1080p picture; 32 threads of it hand over 33 000 pictures/s
// at 44 % of what the PCIe link carries (round 3), against 19 000 real-time streams' worth of reconstruction on the
// device.  In a device-packed stage the host only COPIES the C ABI's arrays as they are — mpeghip_pic_desc,
// mpeghip_mb_desc, the sparse hand-over's count / pair words (or lets the parser write them straight into the pinned
// staging buffer: mpeghip_video_stage_map) — and pack_kernel, in front of recon_kernel on the same stream, turns them into
// chunks and words.
//
// One lane = one macroblock; a wave takes 64 consecutive macroblocks = 16 chunks of one picture, in three phases with a
// wave-private exchange array in LDS between them (no barrier):
//   0  the wave's window: the dwords its macroblocks can name, copied into LDS by the whole wave (PkWin below)
//   1  pk_scan:  validate the lane's descriptor (the checks of validate_mb, one by one), mark its position in the
//                picture's bitmap (a position named twice: the same rule as the host's), work out its record (d0..d3:
//                the window's tile / block offsets, half-pel flags, kRSlow) and walk its coded blocks' count words
//   2  pk_share: every lane publishes its totals (blocks, entries, dwords of snapshot / dense data, where its data ends)
//   3  pk_emit:  from its chunk's four summaries: its slots, where its entries and its other data go; the lane writes its
//                record, its block words, its entries `pair | bits` and its snapshot / dense blocks; the chunk's first
//                lane writes the header
// WHERE a chunk's words go needs no scan over the picture: a sparse picture's macroblocks name their words in order and
// without overlap (mbs[k + 1].coef_off >= the end of macroblock k's data: the ABI's rule, checked here as on the host),
// and the packed form of a block is never longer than its input (count word -> block word, pair -> entry, the DC pair into
// the block word, more than 32 pairs -> a 32-dword unit, a snapshot's count word -> its block word).  So chunk c's words
// start at the dword index of its first macroblock's coef_off, in a words array as long as the input's.
//
// Errors cannot make the call fail — it has returned by the time the device looks at the data.  Every lane that finds
// something wrong reports (global macroblock index, reason) into its PICTURE's 64-bit word by atomic minimum (so a picture's
// report names its first bad macroblock in submit order, as the host's would); pack_gate_kernel, behind pack_kernel on the
// stream, turns every chunk of a refused picture into a dead one — that picture is not reconstructed, the commit's other
// pictures (other streams: pictures of one stream never share a commit) are — and puts the first report, the number of refused
// pictures and their indices where the host finds them at its next synchronisation point (mpeghip_video_verdict / _sync and
// friends).
#pragma once

#include "video_recon_lane.h"

namespace mpg {

// per picture, beside its mpeghip_pic_desc (built by the host at stage_begin, except `use`)
struct PkPic {
    uint32_t word_first; // dword index of the picture's first word: in the staged input AND in the packed words
    uint32_t n_words;    // dwords of its sparse data
    uint32_t chunk_first;
    uint32_t use;        // written by pack_kernel: bit 0 some macroblock predicts from pic.fwd, bit 1 from pic.bwd
};

struct PackArgs {
    const mpeghip_pic_desc *pics;
    PkPic *aux;
    const mpeghip_mb_desc *mbs;
    const uint32_t *words_in;
    uint32_t *chunks;    // out: kRcChunkDwords per chunk
    uint32_t *words_out; // out
    uint32_t *seen;      // [n_pics][seen_stride] dwords, zeroed: one bit per macroblock position
    unsigned long long *err; // [n_pics], per picture: ~0 = nothing wrong; else (its first bad macroblock's index in the submit) << 8 | kPk* reason
    uint32_t n_pics, groups_per_pic, seen_stride;
    uint32_t win_dwords; // the waves' LDS window (PkWin)
    uint32_t mb_w, mb_h, luma_w, chroma_w, luma_bytes, chroma_bytes;
    uint64_t frame_bytes, frame_stride, rgba_stride;
};

// reasons (the host turns them into the codes and texts of validate_mb / rc_pack_picture)
constexpr uint32_t kPkPosition = 1, kPkRefs = 2, kPkCbp = 3, kPkQscale = 4, kPkSameSlot = 5, kPkRange = 6, kPkTwice = 7, kPkSparse = 8,
                   kPkOrder = 9, kPkDepends = 10;
constexpr unsigned long long kPkNoError = ~0ull;

// (one word PER PICTURE: a refusal is the picture's own — the commit's other pictures, which belong to other streams, are
// reconstructed; round 6)
MPG_HD void pk_report(const PackArgs &a, uint32_t pic, uint32_t mb_index, uint32_t reason)
{
    const unsigned long long key = ((unsigned long long)mb_index << 8) | reason;
#if MPG_ON_DEVICE
    atomicMin(a.err + pic, key);
#else
    if (key < a.err[pic])
        a.err[pic] = key;
#endif
}
MPG_HD uint32_t pk_fetch_or(uint32_t *p, uint32_t bits)
{
#if MPG_ON_DEVICE
    return atomicOr(p, bits);
#else
    const uint32_t old = *p;
    *p = old | bits;
    return old;
#endif
}

// The wave's view of its picture's words.  Walking a macroblock's blocks is a chain of dependent reads (every count word tells
// where the next one is) and copying its pairs a loop of reads, one lane each: straight from HBM that is a microsecond per
// step, and a wave of the first version took 150 of them.  So the wave first copies the dwords its 64 macroblocks can name
// — from its first macroblock's offset to the next wave's first (the ABI's order rule makes that one contiguous range) —
// into LDS with whole-wave 16-byte loads, all in flight at once, and reads them from there; what does not fit the window
// (a wave of dense macroblocks: 25 000 dwords) is read from memory as before.
struct PkWin {
    const uint32_t *glob; // the picture's words in memory
    const uint32_t *lds;  // dwords [lo, lo + n) of them
    uint32_t lo, n;
};
MPG_HD uint32_t pk_word(const PkWin &w, uint32_t at)
{
    const uint32_t r = at - w.lo;
    if (r < w.n)
        return w.lds[r];
    return w.glob[at];
}
// The words a wave PRODUCES go the same way, backwards: lanes put block words, entries and units into a second window in LDS —
// the same dwords [lo, lo + n) of the packed array, since a chunk's words start at its first macroblock's offset — and the
// wave writes the window out with whole-wave stores.  (Lane by lane — every lane 4 bytes into a cache line of its own per
// store instruction — the write requests were what the first two versions of the kernel took their 300 us per 64
// pictures for: 78 million of them, against 0.8 million now.)  Only dwords of the wave's own territory — from its first
// macroblock's offset to the next wave's — leave the window; what falls beyond the window is stored directly.
struct PkOut {
    uint32_t *glob; // the packed words in memory (index = dword of the whole array)
    uint32_t *lds;  // dwords [base, base + n) of them
    uint32_t base, n;
};
MPG_HD void pk_put(const PkOut &o, uint32_t at, uint32_t v)
{
    const uint32_t r = at - o.base;
    if (r < o.n)
        o.lds[r] = v;
    else
        o.glob[at] = v;
}
MPG_HD void pk_put16(const PkOut &o, uint32_t at, uint32_t half, uint16_t v) // half-word `half` of dword `at`
{
    const uint32_t r = at - o.base;
    uint16_t *p = reinterpret_cast<uint16_t *>(r < o.n ? o.lds + r : o.glob + at);
    p[half] = v;
}
constexpr uint32_t kPkWinDwords = 4096; // 16 KB: a wave of a typical picture names about 1 000 dwords, of an I picture about 3 200
// the window of the wave whose first macroblock's data begins at `lo` and whose successor's at `hi` (the picture has n_words):
// 16-byte aligned start, at most kPkWinDwords
MPG_HD void pk_window_range(uint32_t lo, uint32_t hi, uint32_t n_words, uint32_t cap, uint32_t &lo4, uint32_t &n)
{
    lo4 = lo & ~3u;
    n = 0;
    if (lo <= hi && hi <= n_words) {
        n = hi - lo4;
        n = n < cap ? n : cap;
    }
}

constexpr uint32_t kPkSparseBlk = 0, kPkDenseBlk = 1, kPkRawBlk = 2;
constexpr int kPkXchDwords = 8; // per lane of the exchange array
// exchange word 0
constexpr uint32_t kPkXAnyRaw = 1u << 8, kPkXAnyDense = 1u << 9, kPkXAnyDc = 1u << 10, kPkXRunOk = 1u << 11, kPkXLive = 1u << 12;

struct PkLane {
    uint32_t live;      // the lane has a macroblock (k < the picture's count)
    uint32_t ok;        // ... and nothing is wrong with it: its words are packed
    uint32_t gi;        // its index in the submit's macroblock array
    uint32_t pic;       // its picture's index in the submit
    uint32_t d[kRcRecDwords]; // its record
    uint32_t cbp, intra, raw, qscale, mb_x, mb_y;
    uint32_t coef_off;  // dwords from the picture's first word
    uint32_t end;       // where its data ends
    uint32_t nb;        // coded blocks
    uint32_t blk[6];    // per block b: coded << 31 | kind << 24 | pairs << 16 | (its count word's offset - coef_off)
    uint32_t ents, def_dw;
    uint32_t flags;     // kPkX*
    uint32_t use;       // which reference it reads: 1 = pic.fwd, 2 = pic.bwd
};

// ---- phase 1
MPG_HD PkLane pk_scan(const PackArgs &a, uint32_t pic, const mpeghip_pic_desc &p, const PkPic &x, uint32_t k, const mpeghip_mb_desc &mb,
                      const PkWin &in)
{
    PkLane L;
    L.live = k < p.mb_count ? 1u : 0u;
    L.ok = 0;
    L.gi = p.mb_first + k;
    L.pic = pic;
    L.d[0] = kRDead;
    L.d[1] = L.d[2] = L.d[3] = L.d[4] = L.d[5] = 0;
    L.cbp = L.intra = L.raw = L.qscale = L.mb_x = L.mb_y = 0;
    L.coef_off = L.end = 0;
    L.nb = L.ents = L.def_dw = L.flags = L.use = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
        L.blk[i] = 0;
    if (!L.live)
        return L;
    const bool intra = (mb.flags & MPEGHIP_MB_INTRA) != 0, raw = (mb.flags & MPEGHIP_MB_COEF_RAW) != 0;
    const uint32_t nref = ((mb.flags & MPEGHIP_MB_REF_FWD) ? 1u : 0u) + ((mb.flags & MPEGHIP_MB_REF_BWD) ? 1u : 0u);
    const uint32_t nb = (uint32_t)__builtin_popcount(mb.cbp & 0x3fu);
    L.flags = kPkXLive;
    L.coef_off = L.end = mb.coef_off;
    // ---- the checks of validate_mb (mpeghip.hip), in its order
    uint32_t reason = 0;
    if (mb.mb_x >= a.mb_w || mb.mb_y >= a.mb_h)
        reason = kPkPosition;
    else if ((intra && nref != 0) || (!intra && nref != 1))
        reason = kPkRefs;
    else if (mb.cbp > 0x3f)
        reason = kPkCbp;
    else if (!raw && nb && (mb.qscale == 0 || mb.qscale > 31))
        reason = kPkQscale;
    const int32_t mvx = mb.mv_x, mvy = mb.mv_y;
    const int32_t cmx = mvx / 2, cmy = mvy / 2; // toward zero, video_noasm.go:35-36
    if (!reason && !intra) {
        const uint32_t ref = (mb.flags & MPEGHIP_MB_REF_BWD) ? p.bwd : p.fwd;
        if (ref == p.cur) {
            reason = kPkSameSlot;
        } else { // extents of the reference's copyBlock reads (video_noasm.go:48-80): [plane start, end of base), else Go panics
            const int64_t cap_y = (int64_t)a.frame_bytes;
            const int64_t cap_c0 = (int64_t)(a.frame_bytes - a.luma_bytes);
            const int64_t cap_c1 = (int64_t)(a.frame_bytes - a.luma_bytes - a.chroma_bytes);
            const int64_t lsi = ((int64_t)((uint32_t)mb.mb_y << 4) + (mvy >> 1)) * a.luma_w + ((uint32_t)mb.mb_x << 4) + (mvx >> 1);
            const int64_t llast = lsi + (int64_t)(15 + (mvy & 1)) * a.luma_w + 15 + (mvx & 1);
            const int64_t csi = ((int64_t)((uint32_t)mb.mb_y << 3) + (cmy >> 1)) * a.chroma_w + ((uint32_t)mb.mb_x << 3) + (cmx >> 1);
            const int64_t clast = csi + (int64_t)(7 + (cmy & 1)) * a.chroma_w + 7 + (cmx & 1);
            if (lsi < 0 || llast >= cap_y || csi < 0 || clast >= cap_c1 || clast >= cap_c0)
                reason = kPkRange;
        }
    }
    if (reason != kPkPosition) { // a position may be named once per picture (macroblocks of a submit run concurrently)
        const uint32_t at = (uint32_t)mb.mb_y * a.mb_w + mb.mb_x;
        const uint32_t bit = 1u << (at & 31);
        if ((pk_fetch_or(a.seen + (size_t)pic * a.seen_stride + (at >> 5), bit) & bit) && !reason)
            reason = kPkTwice;
    }
    if (!reason && mb.coef_off > x.n_words)
        reason = kPkSparse;
    if (reason) {
        pk_report(a, L.pic, L.gi, reason);
        return L;
    }
    L.cbp = mb.cbp;
    L.intra = intra;
    L.raw = raw;
    L.qscale = mb.qscale & 31u;
    L.mb_x = mb.mb_x;
    L.mb_y = mb.mb_y;
    // ---- the record: the host packer's own function (video_recon_lane.h)
    rc_make_record(a.mb_w, a.mb_h, a.luma_w, a.chroma_w, a.luma_bytes, a.frame_stride, mb.mb_x, mb.mb_y, mb.cbp, intra, mvx, mvy,
                   (mb.flags & MPEGHIP_MB_REF_BWD) ? p.bwd : p.fwd, L.d);
    if (!intra || mb.cbp == 0x3f) // (an invalid intra block keeps the old pixels: no whole-row stores for its chunk)
        L.flags |= kPkXRunOk;
    // ---- its coded blocks: count word by count word (each tells where the next one is)
    uint32_t at = mb.coef_off;
    bool bad = false;
#pragma unroll
    for (int b = 0; b < 6; b++) { // (blk[] is indexed by the block number: static register indices)
        if (!(mb.cbp & (0x20u >> b)) || bad)
            continue;
        const uint32_t cnt = at < x.n_words ? pk_word(in, at) : 65u;
        if (cnt > 64 || (raw && cnt != 64) || at + 1 + cnt > x.n_words) { // (at <= n_words <= 2^30: no wrap)
            bad = true;
            continue;
        }
        uint32_t kind = raw ? kPkRawBlk : kPkSparseBlk;
        if (!raw) {
            if (intra && (cnt == 0 || (pk_word(in, at + 1) & 0xfcu))) { // an intra block's DC comes first
                bad = true;
                continue;
            }
            // more than kDenseAbove levels: a unit is the shorter form — if the dense path can take them: every level non-zero
            // (a coded zero level dequantises to +-1, video.go:719-736: only an entry says that) and within its 16-bit steps;
            // an intra block's DC is exempt from both
            if (cnt > kDenseAbove) {
                bool as_unit = true;
                for (uint32_t i = intra ? 1u : 0u; i < cnt && as_unit; i++) {
                    const int32_t level = (int16_t)(pk_word(in, at + 1 + i) >> 16);
                    as_unit = level != 0 && level >= -kRcDenseLevelMax && level <= kRcDenseLevelMax;
                }
                kind = as_unit ? kPkDenseBlk : kPkSparseBlk;
            }
        }
        L.blk[b] = (at - mb.coef_off) | (cnt << 16) | (kind << 24) | (1u << 31);
        if (kind == kPkSparseBlk) {
            L.ents += cnt - (intra ? 1u : 0u);
            L.flags |= intra ? kPkXAnyDc : 0u;
        } else {
            L.def_dw += kind == kPkRawBlk ? 64u : 32u;
            L.flags |= kind == kPkRawBlk ? kPkXAnyRaw : kPkXAnyDense;
        }
        at += 1 + cnt;
    }
    if (bad) {
        pk_report(a, L.pic, L.gi, kPkSparse);
        L.ents = L.def_dw = 0;
        return L;
    }
    L.nb = nb;
    L.end = at;
    L.use = intra ? 0u : ((mb.flags & MPEGHIP_MB_REF_BWD) ? 2u : 1u);
    L.ok = 1;
    return L;
}

// entries of block b of a lane (sparse blocks only)
MPG_HD uint32_t pk_block_entries(const PkLane &L, int b)
{
    const uint32_t w = L.blk[b];
    return (w >> 31) && ((w >> 24) & 3u) == kPkSparseBlk ? ((w >> 16) & 0x7fu) - L.intra : 0u;
}

// ---- phase 2: what the other lanes of the chunk need
MPG_HD void pk_share(uint32_t *xch, int lane, const PkLane &L)
{
    uint32_t *x = xch + lane * kPkXchDwords;
    uint64_t ec = 0; // entries per coded block, 8 bits each, in slot order
    uint32_t i = 0;
#pragma unroll
    for (int b = 0; b < 6; b++)
        if (L.blk[b] >> 31) {
            ec |= (uint64_t)pk_block_entries(L, b) << (8 * i);
            i++;
        }
    x[0] = L.nb | L.flags;
    x[1] = L.ents | (L.def_dw << 16);
    x[2] = L.mb_x | (L.mb_y << 16);
    x[3] = L.coef_off;
    x[4] = L.end;
    x[5] = (uint32_t)ec;
    x[6] = (uint32_t)(ec >> 32);
    x[7] = L.ok;
}

// ---- phase 3.  next_coef_off: coef_off of the macroblock behind lane 63's (the caller loads it; the picture's n_words
// behind its last macroblock)
MPG_HD void pk_emit(const PackArgs &a, const mpeghip_pic_desc &p, const PkPic &x, uint32_t k, int lane, const PkLane &L, const uint32_t *xch,
                    uint32_t next_coef_off, const PkWin &in, const PkOut &out)
{
    const uint32_t k0 = k & ~3u;
    if (k0 >= p.mb_count)
        return; // no such chunk
    const uint32_t m = (uint32_t)lane & 3u;
    const uint32_t *q = xch + ((uint32_t)lane & ~3u) * kPkXchDwords; // the chunk's four summaries
    const uint32_t live = p.mb_count - k0 < (uint32_t)kRcMbs ? p.mb_count - k0 : (uint32_t)kRcMbs;
    // macroblocks name their words in order: mine end where the next one's begin, or before
    if (L.ok) {
        const bool last = k + 1 >= p.mb_count;
        const uint32_t next = last ? x.n_words : (lane < 63 ? xch[(lane + 1) * kPkXchDwords + 3] : next_coef_off);
        if (L.end > next)
            pk_report(a, L.pic, last ? L.gi : L.gi + 1, last ? kPkSparse : kPkOrder);
    }
    // the chunk is packed if all its macroblocks are fine and in order INSIDE it (then its packed words fit between its first
    // macroblock's offset and its last one's end)
    bool chunk_ok = true;
    uint32_t slot_base = 0, n_slots = 0, ent_base = 0, ne = 0, def_base = 0, def_total = 0, any = 0;
    bool run = live == (uint32_t)kRcMbs;
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)kRcMbs; j++) {
        const uint32_t *s = q + j * kPkXchDwords;
        if (j >= live)
            continue;
        chunk_ok = chunk_ok && s[7] != 0 && (j + 1 >= live || s[4] <= s[kPkXchDwords + 3]);
        const uint32_t nbj = s[0] & 0xffu, entsj = s[1] & 0xffffu, defj = s[1] >> 16;
        if (j < m) {
            slot_base += nbj;
            ent_base += entsj;
            def_base += defj;
        }
        n_slots += nbj;
        ne += entsj;
        def_total += defj;
        any |= s[0];
        run = run && (s[0] & kPkXRunOk) && rc_run_follows(a.mb_w, q[2] & 0xffffu, q[2] >> 16, s[2] & 0xffffu, s[2] >> 16, j);
    }
    uint32_t *h = a.chunks + (size_t)(x.chunk_first + (k0 >> 2)) * kRcChunkDwords;
    uint32_t *d = h + kRcHeadDwords + m * kRcRecDwords;
    const bool mine = L.live && chunk_ok;
    d[0] = mine ? L.d[0] : (uint32_t)kRDead;
#pragma unroll
    for (int i = 1; i < kRcRecDwords; i++)
        d[i] = mine ? L.d[i] : 0u;
    const uint32_t W = x.word_first + q[3]; // the chunk's first word: where its first macroblock's data began
    if (m == 0) {
        uint32_t counts = 0;
        if (chunk_ok) { // entries per pass of 8 slots
            uint32_t s = 0;
            for (uint32_t j = 0; j < live; j++) {
                const uint32_t *sj = q + j * kPkXchDwords;
                const uint32_t nbj = sj[0] & 0xffu;
                for (uint32_t i = 0; i < nbj; i++, s++)
                    counts += ((i < 4 ? sj[5] >> (8 * i) : sj[6] >> (8 * (i - 4))) & 0xffu) << (10 * (s >> 3));
            }
        }
        uint32_t hd[kRcHeadDwords];
        rc_make_header_base(a.frame_stride, a.luma_bytes, p.stream, p.cur, (q[2] >> 16) * a.mb_w + (q[2] & 0xffffu), hd);
        const uint64_t wat = (uint64_t)W * 4;
        h[0] = hd[0], h[1] = hd[1];
        h[2] = (uint32_t)wat, h[3] = (uint32_t)(wat >> 32);
        h[4] = chunk_ok ? counts | (run ? kCRun : 0u) | ((p.flags & MPEGHIP_PIC_RGBA) ? kCRgba : 0u) : 0u;
        // (a chunk that is not packed: no blocks, no live macroblock — its records are dead)
        h[5] = chunk_ok ? rc_header_flags(n_slots, live, (any & kPkXAnyRaw) != 0, (any & kPkXAnyDense) != 0, (any & kPkXAnyDc) != 0, p.cur, p.stream)
                        : rc_header_flags(0, 0, false, false, false, p.cur, p.stream);
        h[6] = hd[6], h[7] = hd[7];
    }
    if (!mine || !L.nb)
        return;
    // ---- my blocks' words
    const uint32_t bw = W, e0 = W + n_slots; // (dword indices into the packed words)
    uint32_t ent_at = ent_base, def_at = ne + def_base, stray = 0, s = slot_base;
#pragma unroll
    for (int b = 0; b < 6; b++) {
        const uint32_t w = L.blk[b], cnt = (w >> 16) & 0x7fu, kind = (w >> 24) & 3u;
        if (!(w >> 31))
            continue;
        const uint32_t pr = L.coef_off + (w & 0xffffu) + 1; // (index of the block's first pair in the picture's words)
        uint32_t word = (rc_tile_offset(b, 0, m) >> 3) | (b >= 4 ? kBChroma : 0u) | (kind == kPkRawBlk ? kBRaw : 0u);
        if (kind == kPkRawBlk) {
            word |= def_at << 12;
            for (uint32_t t = 0; t < 64; t++)
                pk_put(out, e0 + def_at + t, pk_word(in, pr + t));
            def_at += 64;
        } else if (kind == kPkDenseBlk) { // the block as a unit: 64 int16 levels by position
            word |= kBDense | (L.qscale << 26) | (L.intra ? 0u : 1u << 31) | (def_at << 12);
            for (uint32_t t = 0; t < 32; t++)
                pk_put(out, e0 + def_at + t, 0u);
            for (uint32_t t = 0; t < cnt; t++) {
                const uint32_t pair = pk_word(in, pr + t), pos = (pair >> 2) & 63u;
                stray |= pair;
                pk_put16(out, e0 + def_at + (pos >> 1), pos & 1u, (uint16_t)(pair >> 16));
            }
            def_at += 32;
        } else {
            uint32_t t = 0;
            if (L.intra) { // the DC pair comes first; it rides in the block word
                const uint32_t dc = pk_word(in, pr);
                stray |= dc;
                word |= kBDcWord | ((dc >> 16) << 12);
                t = 1;
            }
            const uint32_t bits = (L.qscale << 11) | (L.intra ? 0u : kENonIntra) | ((s & 7u) << 8);
            for (; t < cnt; t++) {
                const uint32_t pair = pk_word(in, pr + t);
                stray |= pair;
                pk_put(out, e0 + ent_at++, pair | bits);
            }
        }
        pk_put(out, bw + s++, word);
    }
    if (stray & 0xff03u) // bits outside a pair's two fields
        pk_report(a, L.pic, L.gi, kPkSparse);
}

} // namespace mpg
