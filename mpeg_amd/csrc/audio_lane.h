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
// Data flow per super-step of 32 sub-blocks (both channels; 4 waves per workgroup):
//   phase D: wave 0 runs 64 32-point DCTs, one per lane (channel x 32 sub-blocks), entirely in
//            registers, and writes the 64 mirrored outputs of each into a time-indexed V history in
//            LDS: a ring of 48 slots, slot = [half][channel][32] (+1 pad).  The reference's
//            1024-entry ring only ever holds the last 16 slots.
//   phase W: a wave owns ONE sub-block at a time, lane = channel*32 + sample.  Which history slots
//            and window segments the 16 taps read, and in which order, depends only on the ring
//            position at that sub-block (16 cases), which is wave-uniform: the wave branches once
//            to a fully unrolled variant whose LDS reads are "per-lane base + immediate".  To keep
//            the offsets immediate across the ring wrap, the first 15 slots are mirrored behind the
//            ring (slots 48..62), so "the 16 slots ending at T" are always contiguous.
//
// Time slicing: a sub-block depends on the previous 15 only through the V history, and
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
};

constexpr int kAudioWaves = 4;
constexpr int kAudioThreads = 64 * kAudioWaves;
constexpr int kStep = 32;                             // sub-blocks per super-step: 64 DCTs = one full wave
constexpr int kRingSlots = 48;                        // >= kStep + 15 (the window reaches 15 sub-blocks back)
constexpr int kMirrorSlots = 15;                      // ring slots 0..14 are repeated behind the ring
constexpr int kSlotStride = 129;                      // floats per slot: [half 2][channel 2][32] + 1 pad
constexpr int kHistFloats = (kRingSlots + kMirrorSlots) * kSlotStride;
constexpr int kWinFloats = 1024;                      // window as [segment 16][channel 2][32]
constexpr int kAudioLdsFloats = kHistFloats + kWinFloats;
constexpr int kT0 = 16;                               // local time of the launch's first sub-block

// position of V entry x (0..63) inside a history slot, before the channel offset
MPG_HD constexpr int hx(int x) { return (x >> 5) * 64 + (x & 31); }

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

// scatter with the mirror / sign pattern of audio.go:708-771: X[2k] = e[k], X[2k+1] = o[k]
MPG_HD void scatter(const float (&e)[16], const float (&o)[16], float *v)
{
#pragma unroll
    for (int k = 0; k <= 31; k++) {
        const float X = (k & 1) ? o[k >> 1] : e[k >> 1];
        if (k <= 16)
            v[hx(48 - k)] = -X;
        if (k >= 1 && k <= 15)
            v[hx(48 + k)] = -X;
        if (k >= 17) {
            v[hx(48 - k)] = -X;
            v[hx(k - 16)] = X;
        }
        if (k == 16)
            v[hx(0)] = X;
    }
    v[hx(16)] = 0.0f;
}

// idct36 for one (channel, sub-block): s = 32 sub-band samples, v = the history slot (at this
// channel's offset) that receives d[dp+0 .. dp+63]; v2 = its mirror or nullptr.
MPG_HD void matrixing(const int32_t *s, float *v, float *v2)
{
    float e[16], o[16];
    int32_t in[32];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const i32x4 g = reinterpret_cast<const i32x4 *>(s)[q];
        in[4 * q + 0] = g.v[0];
        in[4 * q + 1] = g.v[1];
        in[4 * q + 2] = g.v[2];
        in[4 * q + 3] = g.v[3];
    }
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
    scatter(e, o, v);
    if (v2)
        scatter(e, o, v2);
}

template <bool kFma> MPG_HD float tap(float acc, float d, float v)
{
    return kFma ? __builtin_fmaf(d, v, acc) : acc + d * v;
}

// ring position of the slot written at local time T (Audio.vPos after that sub-block, audio.go:383)
MPG_HD int32_t vpos_at(int32_t vpos0, int32_t T) { return (vpos0 - 64 * (T - kT0 + 1)) & 1023; }

MPG_HD int ring_slot(int32_t T) { return (int)((uint32_t)T % (uint32_t)kRingSlots); }
MPG_HD int hist_index(int32_t T, int ch, int x) { return ring_slot(T) * kSlotStride + ch * 32 + hx(x); }

// window table -> LDS, repeated per channel so that a tap's address is segment*64 + lane
MPG_HD void audio_load_window(const AudioArgs &a, int tid, float *lds)
{
    for (int idx = tid; idx < kWinFloats; idx += kAudioThreads)
        lds[kHistFloats + idx] = a.window[(idx >> 6) * 32 + (idx & 31)];
}

// ---- state in: Audio.v ring -> time-indexed history
MPG_HD void audio_load_state(const AudioArgs &a, uint32_t stream, int32_t vpos0, int tid, float *lds)
{
    const float *ring = a.ring + (uint64_t)stream * 2048;
    for (int idx = tid; idx < 2048; idx += kAudioThreads) {
        const int ch = idx >> 10, ra = idx & 1023;
        const int e = (ra - vpos0) & 1023;           // slot vpos0 holds the newest block (time T0-1)
        const int T = kT0 - 1 - (e >> 6);            // 0..15: no mirror needed, the first reader has T >= 16
        lds[hist_index(T, ch, e & 63)] = ring[idx];
    }
    audio_load_window(a, tid, lds);
}

// one DCT: sub-block tg (counted from the launch's first) of channel ch -> its history slot (+ mirror)
MPG_HD void hist_matrixing(const AudioArgs &a, uint32_t stream, uint32_t tg, int ch, float *lds)
{
    const uint32_t f = tg / 36, t = tg % 36;
    const int32_t T = kT0 + (int32_t)tg;
    const int32_t slot = ring_slot(T);
    const int32_t *s = a.samples + (((uint64_t)stream * a.n_frames + f) * 2 + (uint32_t)ch) * 1152 + t * 32;
    float *v = lds + slot * kSlotStride + ch * 32;
    matrixing(s, v, slot < kMirrorSlots ? v + kRingSlots * kSlotStride : nullptr);
}

// ---- history rebuild for a slice that starts at frame f0 > 0: the 15 sub-blocks before it
MPG_HD void audio_phase_warmup(const AudioArgs &a, uint32_t stream, uint32_t f0, int tid, float *lds)
{
    if (tid >= 30)
        return;
    hist_matrixing(a, stream, f0 * 36 - 15 + (uint32_t)(tid % 15), tid / 15, lds);
}

// frames [f0, f1) of time slice `chunk`
MPG_HD void audio_chunk_range(const AudioArgs &a, uint32_t chunk, uint32_t &f0, uint32_t &f1)
{
    const uint32_t per = (a.n_frames + a.n_chunks - 1) / a.n_chunks;
    f0 = chunk * per < a.n_frames ? chunk * per : a.n_frames;
    f1 = f0 + per < a.n_frames ? f0 + per : a.n_frames;
}

// ---- phase D: lane (channel, j) of ONE wave transforms sub-block base + j.  The wave rotates from
// step to step so that the workgroups resident on a CU do not all load the same SIMD.
MPG_HD void audio_phase_dct(const AudioArgs &a, uint32_t stream, uint32_t base, uint32_t tg1, int tid, float *lds)
{
    if ((uint32_t)(tid >> 6) != (base / kStep) % kAudioWaves)
        return;
    const uint32_t tg = base + (uint32_t)(tid & 31);
    if (tg < tg1)
        hist_matrixing(a, stream, tg, (tid >> 5) & 1, lds);
}

// The 16 taps of one output sample when the ring position is 64*M (audio_noasm.go:8-38).
// vb = &history[(top - 15) * kSlotStride + lane] where `top` is the (possibly mirrored) slot of
// this sub-block; db = &window_lds[lane].  Everything else folds to immediates.
template <int M, bool kFma> MPG_HD float window_taps(const float *vb, const float *db)
{
    constexpr int32_t pos = 64 * M;
    constexpr int32_t v0 = (pos & 127) >> 1;
    constexpr int32_t d0 = 512 - (pos >> 1);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) { // audio_noasm.go:14-24 — first run of 8 taps
        const int32_t e = (v0 - pos + 128 * k) & 1023;
        const int32_t seg = ((d0 + 64 * k) & 511) >> 5;
        acc = tap<kFma>(acc, db[seg * 64], vb[(15 - (e >> 6)) * kSlotStride + ((e & 63) >> 5) * 64]);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) { // audio_noasm.go:26-37 — second run
        const int32_t e = (96 - v0 - pos + 128 * k) & 1023;
        const int32_t seg = ((d0 + 32 + 64 * k) & 511) >> 5;
        acc = tap<kFma>(acc, db[seg * 64], vb[(15 - (e >> 6)) * kSlotStride + ((e & 63) >> 5) * 64]);
    }
    return acc;
}

template <bool kFma> MPG_HD float window_dispatch(int32_t m, const float *vb, const float *db)
{
    switch (m) {
    case 0: return window_taps<0, kFma>(vb, db);
    case 1: return window_taps<1, kFma>(vb, db);
    case 2: return window_taps<2, kFma>(vb, db);
    case 3: return window_taps<3, kFma>(vb, db);
    case 4: return window_taps<4, kFma>(vb, db);
    case 5: return window_taps<5, kFma>(vb, db);
    case 6: return window_taps<6, kFma>(vb, db);
    case 7: return window_taps<7, kFma>(vb, db);
    case 8: return window_taps<8, kFma>(vb, db);
    case 9: return window_taps<9, kFma>(vb, db);
    case 10: return window_taps<10, kFma>(vb, db);
    case 11: return window_taps<11, kFma>(vb, db);
    case 12: return window_taps<12, kFma>(vb, db);
    case 13: return window_taps<13, kFma>(vb, db);
    case 14: return window_taps<14, kFma>(vb, db);
    default: return window_taps<15, kFma>(vb, db);
    }
}

// ---- phase W: wave w takes sub-blocks base + w + 4n (n = 0..7); lane = channel*32 + sample
template <bool kFma>
MPG_HD void audio_phase_window(const AudioArgs &a, uint32_t stream, int32_t vpos0, uint32_t base, uint32_t tg1, int tid,
                               const float *lds)
{
    const int wave = uniform(tid >> 6), lane = tid & 63;
    const int ch = lane >> 5, i = lane & 31;
    const float *db = lds + kHistFloats + lane;
    for (int n = 0; n < kStep / kAudioWaves; n++) {
        const uint32_t tg = base + (uint32_t)(wave + kAudioWaves * n); // wave-uniform from here down to the taps
        if (tg >= tg1)
            break;
        const int32_t T = kT0 + (int32_t)tg;
        const int32_t m = vpos_at(vpos0, T) >> 6;
        const int32_t slot = ring_slot(T);
        const int32_t top = slot < kMirrorSlots ? slot + kRingSlots : slot;
        const float acc = window_dispatch<kFma>(m, lds + (top - 15) * kSlotStride + lane, db);
        const float sv = acc / -1090519040.0f; // audio.go:390
        const uint32_t f = tg / 36, t = tg % 36;
        const uint64_t fb = ((uint64_t)stream * a.n_frames + f) * 2304;
        const uint32_t o = t * 32 + (uint32_t)i;
        switch (a.format) {
        case MPEGHIP_AUDIO_F32N:
            reinterpret_cast<float *>(a.out)[fb + 2 * o + (uint32_t)ch] = sv;
            break;
        case MPEGHIP_AUDIO_F32NLR:
            reinterpret_cast<float *>(a.out)[fb + (uint32_t)ch * 1152 + o] = sv;
            break;
        case MPEGHIP_AUDIO_S16: // audio.go:400-408
            reinterpret_cast<int16_t *>(a.out)[fb + 2 * o + (uint32_t)ch] = (int16_t)(int32_t)(sv < 0 ? sv * 32768.0f : sv * 32767.0f);
            break;
        default: // MPEGHIP_AUDIO_F32, audio.go:409-417 (both constants are 2^31 in float32)
            reinterpret_cast<float *>(a.out)[fb + 2 * o + (uint32_t)ch] = sv * 2147483648.0f;
            break;
        }
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
        const int T = Tend - 1 - (e >> 6);
        ring[idx] = lds[hist_index(T, ch, e & 63)];
    }
}

MPG_HD void audio_store_vpos(const AudioArgs &a, uint32_t stream, int32_t vpos0)
{
    const int32_t Tend = kT0 + (int32_t)a.n_frames * 36;
    a.vpos_out[stream] = vpos_at(vpos0, Tend - 1);
}

} // namespace mpg
