// audio_lane.h — MP2 sub-band synthesis: one workgroup owns one stream.
//
// Replaces the synthesis loop of Audio.decodeFrame (audio.go:378-422):
//   idct36      audio.go:492-772  (the 32-point "matrixing" DCT)
//   synthWindow audio_noasm.go:8-38 / audio_amd64.s:33-156 / audio_arm64.s:36-85
//   scaling     audio.go:386-418
//
// float32 exactness: every multiply and add below rounds once (the translation
// unit is built with -ffp-contract=off); the window uses fmaf only in
// MPEGHIP_AUDIO_FMA_WINDOW mode, taps are accumulated in the reference's ring
// order (which depends on vPos), the output is a true IEEE division.
//
// Data flow (4 waves per workgroup, steps of 32 sub-blocks, both channels):
//   DCT:    one wave runs the step's 64 32-point DCTs, one per lane (channel x 32 sub-blocks),
//           entirely in registers.  idct36's 64 outputs d[0..63] are a signed mirror of the 32 DCT
//           outputs X[k] (audio.go:708-771: d[48-k] = d[48+k] = -X[k], d[k-16] = X[k], d[16] = 0),
//           so only X is kept: a time-indexed history ring in LDS, slot = [channel][32] (+1 pad).  The
//           reference's 1024-entry ring only ever holds the last 16 slots.
//   window: lane = channel*32 + sample.  A sub-block's 16 taps read the 16 newest history slots, one value each;
//           which slot, which of the lane's two values of it (d[0..31] / d[32..63] half), which window segment and in
//           which ORDER (the float sum depends on it) is a function of the ring position alone, which falls by one per
//           sub-block through a cycle of 16.  Steps start where the position is 15 (audio_step_base0), so the position
//           of every sub-block is a compile-time constant of its index in the step, and a wave that works through
//           CONSECUTIVE sub-blocks keeps the 16 slots in a sliding register file with static indices: two LDS reads
//           per sub-block (its newest slot, both halves) instead of sixteen, addressed as (the run's first slot) + an
//           immediate offset — per sub-block 32 multiplies and adds, one compare, the three instructions of the division
//           and nothing else on the vector unit.  The lane's 16 window coefficients (mirror's sign folded in) are re-read
//           from LDS at the start of a run.
//   pipeline: the history ring holds 79 slots, so the DCTs of step s+1 never touch a slot the
//           windows of step s read.  Iteration s therefore runs, between two barriers, DCT(s+1) on
//           wave (s+1)%4 and window(s) on the other three waves (runs of 11, 11 and 10 sub-blocks, each behind a
//           15-slot lead-in).  The DCT wave takes its samples from an LDS staging buffer and, as soon as it has
//           them in registers, refills the same buffer with the samples of step s+2 straight from
//           HBM (global_load_lds: every lane overwrites exactly the 8 x 16 bytes it has just read):
//           one barrier per step (ordering LDS only), no wave idles through the DCT or its HBM latency.
//           30 780 bytes of LDS and 71 - 73 VGPRs: 5 workgroups stay resident per CU.
//
// Time slicing: a sub-block depends on the previous 15 only through the history, and
// every history slot is a pure function of one sub-block's samples.  So the frames of one
// launch are split into n_chunks slices per stream, one workgroup each; a slice that does not
// start at frame 0 rebuilds its 15-slot history by re-running the 15 DCTs in front of it
// (bit-identical, 15/36 of a frame of extra DCT work per slice).  That is what fills the GPU
// when there are fewer streams than ~8 per CU (BASELINE config 4 has 256 streams).
#pragma once

#include "lane_common.h"
#include "mpeghip.h"

namespace mpg {

struct AudioArgs {
    const int32_t *samples; // [n_streams][n_frames][2][36][32]
    void *out;              // [n_streams][n_frames][2304] of the format's type
    const float *ring;      // [n_streams][2][1024]  (Audio.v) state before this launch
    const int32_t *vpos;    // [n_streams]           (Audio.vPos)
    float *ring_out;        // state after this launch (a different buffer: several workgroups of one
    int32_t *vpos_out;      // stream read the old state while the last one writes the new one)
    uint32_t n_chunks;      // the launch's frames are split into this many time slices per stream
    const float *window;    // [512]                 (synthesisWindow, audio.go:812-899)
    uint32_t n_streams, n_frames;
    int32_t format, fma;
    const uint8_t *active;  // [n_streams] or nullptr: streams with 0 sit this launch out (state carried over unchanged)
};

constexpr int kAudioWaves = 4;
constexpr int kAudioThreads = 64 * kAudioWaves;
constexpr int kSlotStride = 65;                       // floats per slot: [channel 2][32] + 1 pad
constexpr int kT0 = 16;                               // local time of the launch's first sub-block

constexpr int kStep = 32;                             // sub-blocks per step: 64 DCTs = one full wave
constexpr int kRing = 2 * kStep + 15;                 // DCT(s+1) writes [b+32, b+64) while window(s) reads [b-15, b+32)
constexpr int kStageFloats = 64 * 32;                 // one step's samples: [8 x 16 bytes][64 lanes]
constexpr int kHistBase = kStageFloats;               // LDS: the staging buffer, then the history
constexpr int kWinBase = kHistBase + kRing * kSlotStride; // LDS: then the 16 x 32 window coefficients with the mirror's sign
constexpr int kAudioLdsFloats = kWinBase + 512;           // 30780 bytes: 5 workgroups per CU (a CU hands out 160 000)

// c_N[i] = 0.5 / cos((2i+1)*pi/(2N)); identical float32 values to the decimal
// literals of audio.go:498-661.
template <int N> struct DctCoef;
template <> struct DctCoef<32> { static constexpr float c[16] = {
    0.50060299823519630f, 0.50547095989754365f, 0.51544730992262455f, 0.53104259108978417f,
    0.55310389603444452f, 0.58293496820613389f, 0.62250412303566482f, 0.67480834145500568f,
    0.74453627100229858f, 0.83934964541552681f, 0.97256823786196078f, 1.16943993343288470f,
    1.48416461631416620f, 2.05778100995341100f, 3.40760841846871900f, 10.19000812354803300f}; };
template <> struct DctCoef<16> { static constexpr float c[8] = {
    0.50241928618815568f, 0.52249861493968885f, 0.56694403481635769f, 0.64682178335999008f,
    0.78815462345125020f, 1.06067768599034740f, 1.72244709823833420f, 5.10114861868915500f}; };
template <> struct DctCoef<8> { static constexpr float c[4] = {
    0.50979557910415918f, 0.60134488693504529f, 0.89997622313641557f, 2.56291544774150550f}; };
template <> struct DctCoef<4> { static constexpr float c[2] = {0.54119610014619701f, 1.30656296487637640f}; };
template <> struct DctCoef<2> { static constexpr float c[1] = {0.70710678118654746f}; };

// The butterfly network of audio.go:530-706 is this recursion, fully unrolled:
//   e[i] = x[i] + x[N-1-i]      o[i] = (x[i] - x[N-1-i]) * c_N[i]
//   E = dct(e)  O = dct(o)  O[k] += O[k+1] (k ascending)  X[2k] = E[k]  X[2k+1] = O[k]
template <int N>
struct Dct {
    static MPG_HDM void run(float (&x)[N])
    {
        constexpr int H = N / 2;
        float e[H], o[H];
#pragma unroll
        for (int i = 0; i < H; i++) {
            e[i] = x[i] + x[N - 1 - i];
            o[i] = (x[i] - x[N - 1 - i]) * DctCoef<N>::c[i];
        }
        Dct<H>::run(e);
        Dct<H>::run(o);
#pragma unroll
        for (int k = 0; k + 1 < H; k++)
            o[k] += o[k + 1];
#pragma unroll
        for (int k = 0; k < H; k++) {
            x[2 * k] = e[k];
            x[2 * k + 1] = o[k];
        }
    }
};
template <>
struct Dct<1> {
    static MPG_HDM void run(float (&)[1]) {}
};

// the 32 sub-band samples of one (channel, sub-block), 8 x 16 bytes `stride` ints apart
MPG_HD void load_samples(const int32_t *s, int stride, int32_t (&in)[32])
{
#pragma unroll
    for (int q = 0; q < 8; q++) {
        i32x4 g;
        __builtin_memcpy(&g, s + q * stride, 16);
        in[4 * q + 0] = g.v[0];
        in[4 * q + 1] = g.v[1];
        in[4 * q + 2] = g.v[2];
        in[4 * q + 3] = g.v[3];
    }
}

// idct36 for one (channel, sub-block) up to the mirror: v = the history slot (at this channel's offset) that receives
// X[0..31].
MPG_HD void matrixing(const int32_t (&in)[32], float *v)
{
    float e[16], o[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { // audio.go:497-528: integer sum/difference, then float32
        e[i] = (float)(in[i] + in[31 - i]);
        o[i] = (float)(in[i] - in[31 - i]) * DctCoef<32>::c[i];
    }
    Dct<16>::run(e);
    Dct<16>::run(o);
#pragma unroll
    for (int k = 0; k < 15; k++)
        o[k] += o[k + 1];
#pragma unroll
    for (int k = 0; k < 16; k++) { // X[2k] = e[k], X[2k+1] = o[k]
        v[2 * k] = e[k];
        v[2 * k + 1] = o[k];
    }
}

// The mirror (audio.go:708-771): d[x] = sign * X[k]
//   x = 0: X[16]   1..15: X[x+16]   16: 0   17..48: -X[48-x]   49..63: -X[x-48]
MPG_HD int mirror_index(int x) { return x < 16 ? x + 16 : (x == 16 ? 0 : (x <= 48 ? 48 - x : x - 48)); }
MPG_HD float mirror_apply(int x, float X) { return x < 16 ? X : (x == 16 ? 0.0f : -X); }
// and back: X[k] from d
MPG_HD float mirror_recover(int k, const float *d) { return k <= 16 ? -d[48 - k] : d[k - 16]; }

template <bool kFma> MPG_HD float tap(float acc, float d, float v)
{
    return kFma ? __builtin_fmaf(d, v, acc) : acc + d * v;
}

// ring position of the slot written at local time T (Audio.vPos after that sub-block, audio.go:383)
MPG_HD int32_t vpos_at(int32_t vpos0, int32_t T) { return (vpos0 - 64 * (T - kT0 + 1)) & 1023; }

MPG_HD int ring_slot(int32_t T) { return (int)((uint32_t)T % (uint32_t)kRing); }

// ---- state in: Audio.v ring -> time-indexed X history (item = channel, slot time, k)
MPG_HD void audio_load_state(const AudioArgs &a, uint32_t stream, int32_t vpos0, int tid, float *lds)
{
    const float *ring = a.ring + (uint64_t)stream * 2048;
    for (int idx = tid; idx < 1024; idx += kAudioThreads) {
        const int ch = idx >> 9, T = (idx >> 5) & 15, k = idx & 31;
        const int e0 = 64 * (kT0 - 1 - T);                           // slot vpos0 holds the newest block (time T0-1)
        const int x = k <= 16 ? 48 - k : k - 16;
        const float d = ring[ch * 1024 + ((vpos0 + e0 + x) & 1023)];
        lds[kHistBase + ring_slot(T) * kSlotStride + ch * 32 + k] = k <= 16 ? -d : d;
    }
}

// the window coefficients (synthesisWindow, audio.go:812-899) with the mirror's sign, [segment 16][sample 32], into LDS:
// segment s meets d[(s&1)*32 + i].  A wave loads its lanes' 16 values into registers when it starts a run of sub-blocks
// (they are not kept across the DCT, whose 64 registers would otherwise not fit next to them at 5 waves per SIMD).
MPG_HD void audio_store_window(const AudioArgs &a, int tid, float *lds)
{
    for (int idx = tid; idx < 512; idx += kAudioThreads) {
        const int s = idx >> 5, i = idx & 31;
        const float w = a.window[idx];
        lds[kWinBase + idx] = (s & 1) ? -w : mirror_apply(i, w);
    }
}
MPG_HD void audio_load_window(const float *lds, int i, float (&dreg)[16])
{
#pragma unroll
    for (int s = 0; s < 16; s++)
        dreg[s] = lds[kWinBase + s * 32 + i];
}

MPG_HD const int32_t *samples_of(const AudioArgs &a, uint32_t stream, uint32_t tg, int ch)
{
    const uint32_t f = tg / 36, t = tg % 36;
    return a.samples + (((uint64_t)stream * a.n_frames + f) * 2 + (uint32_t)ch) * 1152 + t * 32;
}

// one DCT: sub-block tg (counted from the launch's first) of channel ch -> its history slot (+ repeat)
MPG_HD void hist_matrixing(const int32_t (&in)[32], uint32_t tg, int ch, float *lds)
{
    matrixing(in, lds + kHistBase + ring_slot(kT0 + (int32_t)tg) * kSlotStride + ch * 32);
}

// ---- history rebuild for a slice that starts at sub-block tg0 > 0: the 15 sub-blocks before it
MPG_HD void audio_phase_warmup(const AudioArgs &a, uint32_t stream, uint32_t tg0, int tid, float *lds)
{
    if (tid < 128 || tid >= 128 + 30) // wave 2: waves 0 and 1 issue the first two fetches
        return;
    const int ch = (tid - 128) / 15;
    const uint32_t tg = tg0 - 15 + (uint32_t)((tid - 128) % 15);
    int32_t in[32];
    load_samples(samples_of(a, stream, tg, ch), 4, in);
    hist_matrixing(in, tg, ch, lds);
}

// sub-blocks [tg0, tg1) of time slice `chunk` of a stream whose ring position at the launch's start is vpos0.  Slices
// are cut on the stream's own step grid (audio_step_base0: sub-blocks congruent to vpos0 / 64 modulo 32 here), in whole
// steps: only the launch's first and last step are partly filled, not every slice's.  (Slices are sub-blocks, not frames:
// output element tg * 64 + ... does not care.)
MPG_HD void audio_slice_range(const AudioArgs &a, uint32_t chunk, int32_t vpos0, uint32_t &tg0, uint32_t &tg1)
{
    const uint32_t n = a.n_frames * 36, g = (uint32_t)(vpos0 >> 6) & 15;
    const uint32_t steps = n > g ? (n - g + kStep - 1) / kStep : 0; // from sub-block g on
    auto cut = [&](uint32_t i) -> uint32_t {
        const uint32_t k = i >= a.n_chunks ? steps : (uint32_t)(((uint64_t)i * steps) / a.n_chunks);
        if (i == 0 || k == 0)
            return 0; // (a slice that would end inside the first step is empty: the next one starts at 0)
        if (i >= a.n_chunks)
            return n;
        const uint32_t at = g + kStep * k;
        return at < n ? at : n;
    };
    tg0 = cut(chunk);
    tg1 = cut(chunk + 1);
}

// the wave that runs the DCTs of step `si` (counted from the slice's first): it rotates so that the
// workgroups resident on a CU do not all load the same SIMD
MPG_HD uint32_t dct_wave(uint32_t si) { return si % kAudioWaves; }

// The step grid of a slice.  Steps are 32 sub-blocks long and start where the window's ring position is 15 (it cycles
// through 15 .. 0 as the sub-blocks go by), i.e. at sub-blocks congruent to vpos0 / 64 modulo 16: the position of every
// sub-block of a step is then a compile-time constant of its index in the step (15 - (p & 15)), which is what lets the
// window run straight-line code with a static register file.  The first step of a slice therefore begins up to 15
// sub-blocks BEFORE the slice (base0 <= tg0, possibly negative): those are neither transformed (their history slots hold
// the state / the rebuilt history) nor synthesised.
MPG_HD int32_t audio_step_base0(int32_t vpos0, uint32_t tg0)
{
    const int32_t v = (vpos0 >> 6) & 15;
    return (int32_t)tg0 - (((int32_t)tg0 - v) & 15);
}
MPG_HD uint32_t audio_step_count(int32_t base0, uint32_t tg1) { return (uint32_t)((int32_t)tg1 - base0 + kStep - 1) / kStep; }
MPG_HD bool audio_in_slice(int32_t tg, uint32_t tg0, uint32_t tg1) { return tg >= (int32_t)tg0 && tg < (int32_t)tg1; }

// ---- samples of step si -> the staging buffer, issued by wave si%4 (lane = channel*32 + j); they
// have landed after the next barrier
MPG_HD void audio_phase_fetch(const AudioArgs &a, uint32_t stream, int32_t base0, uint32_t tg0, uint32_t tg1, uint32_t si, int tid,
                              float *lds)
{
    if ((uint32_t)(tid >> 6) != dct_wave(si))
        return;
    const int lane = tid & 63;
    const int32_t tg = base0 + (int32_t)(si * kStep) + (lane & 31);
    if (!audio_in_slice(tg, tg0, tg1))
        return;
    const int32_t *src = samples_of(a, stream, (uint32_t)tg, lane >> 5);
#pragma unroll
    for (int q = 0; q < 8; q++)
        copy16_to_lds(src + 4 * q, lds + q * 256, lane);
}

// ---- DCTs of step si from the staging buffer, on wave si%4; the same lanes refill the buffer for
// step si+1 (whose DCT wave is the next one) once their own samples are in registers
MPG_HD void audio_phase_dct(const AudioArgs &a, uint32_t stream, int32_t base0, uint32_t tg0, uint32_t tg1, uint32_t si, int tid,
                            float *lds)
{
    if ((uint32_t)(tid >> 6) != dct_wave(si))
        return;
    const int lane = tid & 63;
    const int32_t tg = base0 + (int32_t)(si * kStep) + (lane & 31);
    const bool mine = audio_in_slice(tg, tg0, tg1);
    int32_t in[32];
    if (mine)
        load_samples(reinterpret_cast<const int32_t *>(lds) + lane * 4, 256, in);
#if MPG_ON_DEVICE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the reads above are done before the refill is issued
#endif
    const int32_t tn = tg + kStep; // same lane, next step
    if (audio_in_slice(tn, tg0, tg1)) {
        const int32_t *src = samples_of(a, stream, (uint32_t)tn, lane >> 5);
#pragma unroll
        for (int q = 0; q < 8; q++)
            copy16_to_lds(src + 4 * q, lds + q * 256, lane);
    }
    if (mine)
        hist_matrixing(in, (uint32_t)tg, lane >> 5, lds);
}

// The window as a sliding register file.  The 16 taps of a sub-block read the 16 newest history slots, one value each:
// from the slot at distance d (0 = its own) the lane's d[0..31] value (parity 0) or its d[32..63] value (parity 1); which
// distance, which parity and which window segment tap k uses — and the ORDER of the taps, which the float sum depends on —
// is a function of the ring position 64*M alone (audio_noasm.go:8-38).  M falls by one per sub-block while every slot's
// distance grows by one, so the slot at distance d of the sub-block at position M can live in register pair (M + d) mod 16
// for as long as it is needed: a wave that works through CONSECUTIVE sub-blocks reads two new values per sub-block (its
// newest slot, both parities) instead of sixteen, and all register indices are compile-time constants of the variant M.
// (The previous form read 16 values per sub-block through immediate offsets, which needed the first 15 ring slots
// repeated behind the ring: 3.9 KB of LDS that now buys a fifth resident workgroup per CU.)
template <int M, bool kFma> MPG_HD float window_sum(const float (&r0)[16], const float (&r1)[16], const float (&d)[16])
{
    constexpr int32_t pos = 64 * M;
    constexpr int32_t v0 = (pos & 127) >> 1;
    constexpr int32_t d0 = 512 - (pos >> 1);
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) { // audio_noasm.go:14-24 — first run of 8 taps
        const int32_t e = (v0 - pos + 128 * k) & 1023;
        const int32_t seg = ((d0 + 64 * k) & 511) >> 5;
        const int q = (M + (e >> 6)) & 15;
        a = tap<kFma>(a, d[seg], ((e & 63) >> 5) ? r1[q] : r0[q]);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) { // audio_noasm.go:26-37 — second run
        const int32_t e = (96 - v0 - pos + 128 * k) & 1023;
        const int32_t seg = ((d0 + 32 + 64 * k) & 511) >> 5;
        const int q = (M + (e >> 6)) & 15;
        a = tap<kFma>(a, d[seg], ((e & 63) >> 5) ? r1[q] : r0[q]);
    }
    return a;
}

// x / -1090519040 (audio.go:390), correctly rounded.  The short form is Markstein's sequence
// q = x*y, r = fma(-q, D, x), q' = fma(r, y, q) with y = RN(1/D); tests/proofs/div_const.c checks
// all 2^32 inputs: it equals the IEEE quotient for x == 0 and for every finite |x| >= 2^-95 (it
// differs for some 2^-119 <= |x| < 2^-95, where r underflows).  |x| cannot overflow: |samples| <
// 2^31, each of the DCT's 5 levels at most doubles twice and scales by < 10.2 (< 2^27 in all), the
// window sums 16 products with |D| < 2^17: |x| < 2^80.
constexpr float kScale = -1090519040.0f;
constexpr float kScaleInv = 1.0f / kScale;
MPG_HD bool scale_short_ok(float x) { return !(__builtin_fabsf(x) < 0x1p-95f) || x == 0.0f; }
MPG_HD float scale_short(float x)
{
    const float q = x * kScaleInv;
    const float r = __builtin_fmaf(-q, kScale, x);
    return __builtin_fmaf(r, kScaleInv, q);
}

// convert (audio.go:386-418) and store one sample of sub-block tg: the planar and the int16 format.  (The interleaved float
// formats go out from WinSteps::run: frame f, sub-block t, sample i, channel ch sits at element (f*36 + t)*64 + 2i + ch =
// tg*64 + (2i + ch) of the stream's output — no frame / sub-block split, one scalar base per run of sub-blocks.)
// Output samples leave as non-temporal stores: nothing on the device reads them again, and kept out of L2 they leave it
// to the sample loads and the window table (profiles/r8_ab_audio_cache_policy.txt, F32N: +6.5 %; plain stores again at
// 2048 streams, profiles/r23: no better there, 20 % worse on config 4).
template <int kFormat>
MPG_HD void audio_store_sample(const AudioArgs &a, uint32_t stream, uint32_t tg, int ch, int i, float sv)
{
    static_assert(kFormat == MPEGHIP_AUDIO_F32NLR || kFormat == MPEGHIP_AUDIO_S16, "the interleaved float formats are stored by the run");
    const uint64_t sb = (uint64_t)stream * a.n_frames * 2304; // the stream's first output element
    if (kFormat == MPEGHIP_AUDIO_F32NLR) {
        const uint32_t f = tg / 36, t = tg % 36;
        store_streaming(reinterpret_cast<float *>(a.out) + sb + f * 2304 + (uint32_t)ch * 1152 + t * 32 + (uint32_t)i, sv);
        return;
    }
    const uint32_t e = tg * 64 + 2 * (uint32_t)i + (uint32_t)ch; // audio.go:400-408
    store_streaming(reinterpret_cast<int16_t *>(a.out) + sb + e, (int16_t)(int32_t)(sv < 0 ? sv * 32768.0f : sv * 32767.0f));
}

// An LDS location as the window's reads carry it: the 32-bit LDS byte address on the device (so that `address + constant`
// becomes a ds_read_b32 with an immediate offset, whatever the compiler knows about the generic pointer it came from), a
// plain pointer in the emulator.
#if MPG_ON_DEVICE
typedef uint32_t lds_at_t;
MPG_HD lds_at_t lds_at(const float *p) { return (uint32_t)(uintptr_t)p; } // (low 32 bits of an LDS address = its offset)
MPG_HD lds_at_t lds_at_moved(lds_at_t a, int32_t floats) { return a + (uint32_t)(floats * 4); }
template <int kFloats> MPG_HD float lds_read_at(lds_at_t a)
{
    typedef const volatile float __attribute__((address_space(3))) * lds_float_ptr; // (volatile: one ds_read_b32, left where it is)
    return *(lds_float_ptr)(a + (uint32_t)(kFloats * 4));
}
#else
typedef const float *lds_at_t;
MPG_HD lds_at_t lds_at(const float *p) { return p; }
MPG_HD lds_at_t lds_at_moved(lds_at_t a, int32_t floats) { return a + floats; }
template <int kFloats> MPG_HD float lds_read_at(lds_at_t a) { return a[kFloats]; }
#endif

// A run of N consecutive sub-blocks P0 .. P0 + N - 1 of a step (indices in the step: positions and register indices are
// compile-time constants, see audio_step_base0).  Lead-in: the 15 slots in front of the run, sub-block P0 - j into
// register pair (M(P0) + j) & 15.  Then per sub-block: file its newest slot (read one sub-block ahead), synthesise.
//
// The run reads 15 + N CONSECUTIVE ring slots, J = 0 .. 14 + N counted from its oldest.  Their addresses are the lane's two
// addresses of slot J = 0 (computed once per run) plus the compile-time constant J * kSlotStride — the read instruction's
// immediate offset, no vector instruction per read — except that the ring may wrap once inside the run: `wrap_at` is the J
// of the first slot behind the wrap (wave-uniform; >= the run's length if there is none), where both addresses step back by
// one ring, behind a scalar branch.  (Before: a scalar multiply and two vector adds per read, 2 x (15 + N) vector
// instructions per run of N sub-blocks on top of its 39 x N.)
struct WinFile {
    float r0[16], r1[16];
    float n0, n1;      // the newest slot of the sub-block about to be worked on
    lds_at_t at0, at1; // the lane's d[0..31] / d[32..63] value of the run's slot J = 0
    uint32_t wrap_at;
    // wave-uniform, set once per run, so that a sub-block costs two scalar instructions of bookkeeping and not fifteen:
    uint32_t rel0, len; // sub-block P of the step lies inside the slice iff (uint32_t)(rel0 + P) < len
    uint8_t *out0;      // interleaved float formats: where the run's first sub-block goes; sub-block P0 + n: + 256 n bytes
};
template <int J> MPG_HD void audio_window_read(WinFile &w, float &v0, float &v1)
{
    if (__builtin_expect((uint32_t)J == w.wrap_at, 0)) { // (wave-uniform; out of line: the common path falls through)
        w.at0 = lds_at_moved(w.at0, -kRing * kSlotStride);
        w.at1 = lds_at_moved(w.at1, -kRing * kSlotStride);
        sched_fence(); // keeps this a branch around two adds, not two selects on every read
    }
    v0 = lds_read_at<J * kSlotStride>(w.at0);
    v1 = lds_read_at<J * kSlotStride>(w.at1);
}
template <int J, int kLast> struct WinLeadIn { // J = 0 .. 14: sub-block P0 - 15 + J, oldest first
    template <int M0> static MPG_HDM void run(WinFile &w)
    {
        audio_window_read<J>(w, w.r0[(M0 + 15 - J) & 15], w.r1[(M0 + 15 - J) & 15]);
        WinLeadIn<J + 1, kLast>::template run<M0>(w);
    }
};
template <int kLast> struct WinLeadIn<kLast, kLast> {
    template <int M0> static MPG_HDM void run(WinFile &) {}
};
template <int P0, int P, int kEnd, bool kFma, int kFormat> struct WinSteps {
    static MPG_HDM void run(const AudioArgs &a, uint32_t stream, int32_t base, uint32_t tg0, uint32_t tg1, int ch, int i,
                            const float (&dreg)[16], WinFile &w)
    {
        constexpr int M = 15 - (P & 15);
        w.r0[M] = w.n0;
        w.r1[M] = w.n1;
        if (P + 1 < kEnd)
            audio_window_read<16 + P - P0>(w, w.n0, w.n1); // the next sub-block's, on their way while this one is summed
        if (w.rel0 + (uint32_t)P < w.len) { // (wave-uniform) audio_in_slice(base + P, tg0, tg1)
            const float acc = window_sum<M, kFma>(w.r0, w.r1, dreg);
            // One compare decides in the common case; sums that are exactly zero (digital silence) fail it and are let
            // through by the second, exact test.
            float sv;
            if (__builtin_expect(all_in_wave(!(__builtin_fabsf(acc) < 0x1p-95f)), 1) || all_in_wave(scale_short_ok(acc)))
                sv = scale_short(acc);
            else
                sv = acc / kScale;
            if (kFormat == MPEGHIP_AUDIO_F32N || kFormat == MPEGHIP_AUDIO_F32) // audio.go:409-417 (both constants are 2^31 in float32)
                store32_streaming_at_imm<(P - P0) * 256>(w.out0, (2 * (uint32_t)i + (uint32_t)ch) * 4,
                                                         kFormat == MPEGHIP_AUDIO_F32N ? sv : sv * 2147483648.0f);
            else
                audio_store_sample<(kFormat == MPEGHIP_AUDIO_F32NLR || kFormat == MPEGHIP_AUDIO_S16) ? kFormat : MPEGHIP_AUDIO_S16>(
                    a, stream, (uint32_t)(base + P), ch, i, sv);
        }
        WinSteps<P0, P + 1, kEnd, kFma, kFormat>::run(a, stream, base, tg0, tg1, ch, i, dreg, w);
    }
};
template <int P0, int kEnd, bool kFma, int kFormat> struct WinSteps<P0, kEnd, kEnd, kFma, kFormat> {
    static MPG_HDM void run(const AudioArgs &, uint32_t, int32_t, uint32_t, uint32_t, int, int, const float (&)[16], WinFile &) {}
};
template <int P0, int kEnd, bool kFma, int kFormat>
MPG_HD void audio_window_run(const AudioArgs &a, uint32_t stream, int32_t base, uint32_t tg0, uint32_t tg1, int ch, int i,
                             const float *p0, const float *p1, const float *lds)
{
    if (base + P0 >= (int32_t)tg1 || base + kEnd <= (int32_t)tg0)
        return; // (nothing of this run lies inside the slice)
    float dreg[16];
    audio_load_window(lds, i, dreg);
    WinFile w;
    const uint32_t slot0 = (uint32_t)uniform(ring_slot(kT0 + base + P0 - 15 + kRing)); // (+ kRing: the argument stays positive)
    w.at0 = lds_at(p0 + slot0 * kSlotStride);
    w.at1 = lds_at(p1 + slot0 * kSlotStride);
    w.wrap_at = (uint32_t)kRing - slot0; // 1 .. kRing
    w.rel0 = (uint32_t)uniform((int32_t)((uint32_t)base - tg0));
    w.len = (uint32_t)uniform((int32_t)(tg1 - tg0));
    // (base + P0 may lie in front of the slice, even of the stream's output: only sub-blocks inside the slice are stored)
    w.out0 = reinterpret_cast<uint8_t *>(a.out) + ((int64_t)((uint64_t)stream * a.n_frames * 2304) + (int64_t)(base + P0) * 64) * 4;
    constexpr int M0 = 15 - (P0 & 15);
    WinLeadIn<0, 15>::template run<M0>(w);
    audio_window_read<15>(w, w.n0, w.n1);
    WinSteps<P0, P0, kEnd, kFma, kFormat>::run(a, stream, base, tg0, tg1, ch, i, dreg, w);
}

// ---- windows of step si: the three waves that do not run DCT(si + 1) take sub-blocks 0..10, 11..21 and 22..31.
template <bool kFma, int kFormat>
MPG_HD void audio_phase_window(const AudioArgs &a, uint32_t stream, int32_t vpos0, int32_t base0, uint32_t tg0, uint32_t tg1,
                               uint32_t si, int tid, const float *lds)
{
    const uint32_t wave = (uint32_t)uniform(tid >> 6), busy = dct_wave(si + 1);
    const uint32_t rank = (wave - busy - 1) % kAudioWaves; // 0..2 for the three free waves, 3 for the busy one
    const int lane = tid & 63, ch = lane >> 5, i = lane & 31;
    const float *p0 = lds + kHistBase + ch * 32 + mirror_index(i);
    const float *p1 = lds + kHistBase + ch * 32 + mirror_index(32 + i);
    const int32_t base = base0 + (int32_t)(si * kStep);
    MPG_CHECK((vpos_at(vpos0, kT0 + base) >> 6) == 15); // the grid is aligned to the position cycle
    (void)vpos0;
    // the free waves take [0, a), [a, b), [b, c), the DCT wave [c, 32): 10 / 10 / 10 / 2 and 10 / 11 / 10 / 1 measure the
    // same (profiles/r3y_ab_audio_run_splits.txt)
    constexpr int kCut[3] = {11, 22, 32};
    switch (rank) {
    case 0: audio_window_run<0, kCut[0], kFma, kFormat>(a, stream, base, tg0, tg1, ch, i, p0, p1, lds); break;
    case 1: audio_window_run<kCut[0], kCut[1], kFma, kFormat>(a, stream, base, tg0, tg1, ch, i, p0, p1, lds); break;
    case 2: audio_window_run<kCut[1], kCut[2], kFma, kFormat>(a, stream, base, tg0, tg1, ch, i, p0, p1, lds); break;
    default:
        if (kCut[2] < kStep)
            audio_window_run<kCut[2], kStep, kFma, kFormat>(a, stream, base, tg0, tg1, ch, i, p0, p1, lds);
        break;
    }
}

// ---- state out: last 16 history slots -> Audio.v ring; thread 0 advances vPos
MPG_HD void audio_store_state(const AudioArgs &a, uint32_t stream, int32_t vpos0, int tid, const float *lds)
{
    float *ring = a.ring_out + (uint64_t)stream * 2048;
    const int32_t Tend = kT0 + (int32_t)a.n_frames * 36; // first time NOT produced
    const int32_t vpos1 = vpos_at(vpos0, Tend - 1);
    for (int idx = tid; idx < 2048; idx += kAudioThreads) {
        const int ch = idx >> 10, ra = idx & 1023;
        const int e = (ra - vpos1) & 1023;
        const int T = Tend - 1 - (e >> 6), x = e & 63;
        ring[idx] = mirror_apply(x, lds[kHistBase + ring_slot(T) * kSlotStride + ch * 32 + mirror_index(x)]);
    }
}

// ---- a stream that sits the launch out: its state moves to the new buffers untouched
MPG_HD void audio_carry_state(const AudioArgs &a, uint32_t stream, int tid)
{
    for (int idx = tid; idx < 2048; idx += kAudioThreads)
        a.ring_out[(uint64_t)stream * 2048 + idx] = a.ring[(uint64_t)stream * 2048 + idx];
    if (tid == 0)
        a.vpos_out[stream] = a.vpos[stream];
}

MPG_HD void audio_store_vpos(const AudioArgs &a, uint32_t stream, int32_t vpos0)
{
    const int32_t Tend = kT0 + (int32_t)a.n_frames * 36;
    a.vpos_out[stream] = vpos_at(vpos0, Tend - 1);
}

} // namespace mpg
