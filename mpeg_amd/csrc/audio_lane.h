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
// Data flow per frame (36 sub-blocks x 2 channels):
//   phase D: 72 threads run one 32-point DCT each, entirely in registers, and
//            write the 64 mirrored outputs into a time-indexed V history in LDS
//            (64 slots of 64 floats per channel, padded to 65: the reference's
//            1024-entry ring only ever holds the last 16 slots).
//   phase W: 1152 (sub-block, sample) pairs over 384 threads, 3 each, both
//            channels per thread: 16 taps per channel from LDS, scale, store L/R.
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

constexpr int kAudioThreads = 384;
constexpr int kHistSlots = 64;
constexpr int kHistStride = 65;                       // floats per slot (64 + 1 pad)
constexpr int kHistFloats = 2 * kHistSlots * kHistStride;
constexpr int kAudioLdsFloats = kHistFloats + 512;    // + window table
constexpr int kT0 = 16;                               // local time of the first new sub-block

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

// idct36 for one (channel, sub-block): s = 32 sub-band samples, v = the 64-float
// history slot that receives d[dp+0 .. dp+63] (audio.go:708-771).
MPG_HD void matrixing(const int32_t *s, float *v)
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
    // X[2k] = e[k], X[2k+1] = o[k]; scatter with the mirror / sign pattern
#pragma unroll
    for (int k = 0; k <= 31; k++) {
        const float X = (k & 1) ? o[k >> 1] : e[k >> 1];
        if (k <= 16)
            v[48 - k] = -X;
        if (k >= 1 && k <= 15)
            v[48 + k] = -X;
        if (k >= 17) {
            v[48 - k] = -X;
            v[k - 16] = X;
        }
        if (k == 16)
            v[0] = X;
    }
    v[16] = 0.0f;
}

MPG_HD float tap(float acc, float d, float v, bool fma)
{
    return fma ? __builtin_fmaf(d, v, acc) : acc + d * v;
}

// ring position of the slot written at local time T
MPG_HD int32_t vpos_at(int32_t vpos0, int32_t T) { return (vpos0 - 64 * (T - kT0 + 1)) & 1023; }

// ---- state in: Audio.v ring -> time-indexed history; window table -> LDS
MPG_HD void audio_load_state(const AudioArgs &a, uint32_t stream, int32_t vpos0, int tid, float *lds)
{
    const float *ring = a.ring + (uint64_t)stream * 2048;
    for (int idx = tid; idx < 2048; idx += kAudioThreads) {
        const int ch = idx >> 10, ra = idx & 1023;
        const int e = (ra - vpos0) & 1023;           // slot vpos0 holds the newest block (time T0-1)
        const int T = kT0 - 1 - (e >> 6);
        lds[(ch * kHistSlots + (T & (kHistSlots - 1))) * kHistStride + (e & 63)] = ring[idx];
    }
    for (int idx = tid; idx < 512; idx += kAudioThreads)
        lds[kHistFloats + idx] = a.window[idx];
}

// ---- history rebuild for a slice that starts at frame f0 > 0: the 15 sub-blocks before it
MPG_HD void audio_phase_warmup(const AudioArgs &a, uint32_t stream, uint32_t f0, int tid, float *lds)
{
    if (tid >= 30)
        return;
    const int ch = tid / 15, t = 21 + tid % 15;
    const uint32_t f = f0 - 1;
    const int32_t T = kT0 + (int32_t)f * 36 + t;
    const int32_t *s = a.samples + (((uint64_t)stream * a.n_frames + f) * 2 + (uint32_t)ch) * 1152 + (uint32_t)t * 32;
    matrixing(s, lds + (ch * kHistSlots + (T & (kHistSlots - 1))) * kHistStride);
}

MPG_HD void audio_load_window(const AudioArgs &a, int tid, float *lds)
{
    for (int idx = tid; idx < 512; idx += kAudioThreads)
        lds[kHistFloats + idx] = a.window[idx];
}

// frames [f0, f1) of time slice `chunk`
MPG_HD void audio_chunk_range(const AudioArgs &a, uint32_t chunk, uint32_t &f0, uint32_t &f1)
{
    const uint32_t per = (a.n_frames + a.n_chunks - 1) / a.n_chunks;
    f0 = chunk * per < a.n_frames ? chunk * per : a.n_frames;
    f1 = f0 + per < a.n_frames ? f0 + per : a.n_frames;
}

// ---- phase D: thread `tid` < 72 transforms (ch, t) of frame f
MPG_HD void audio_phase_dct(const AudioArgs &a, uint32_t stream, uint32_t f, int tid, float *lds)
{
    if (tid >= 72)
        return;
    const int ch = tid / 36, t = tid % 36;
    const int32_t T = kT0 + (int32_t)f * 36 + t;
    const int32_t *s = a.samples + (((uint64_t)stream * a.n_frames + f) * 2 + (uint32_t)ch) * 1152 + (uint32_t)t * 32;
    matrixing(s, lds + (ch * kHistSlots + (T & (kHistSlots - 1))) * kHistStride);
}

// ---- phase W: thread handles pairs p = tid + 384*n (n = 0..2): t = p>>5, i = p&31
MPG_HD void audio_phase_window(const AudioArgs &a, uint32_t stream, int32_t vpos0, uint32_t f, int tid, const float *lds)
{
    const float *dtab = lds + kHistFloats;
    const bool fma = a.fma != 0;
    for (int n = 0; n < 3; n++) {
        const int p = tid + kAudioThreads * n;
        const int t = p >> 5, i = p & 31;
        const int32_t T = kT0 + (int32_t)f * 36 + t;
        const int32_t pos = vpos_at(vpos0, T);
        const int32_t v0 = (pos & 127) >> 1;
        const int32_t d0 = 512 - (pos >> 1);
        float accL = 0.0f, accR = 0.0f;
        // audio_noasm.go:14-24 — first run of 8 taps
        int32_t e = (v0 - pos) & 1023, di = d0 + i;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int slot = ((T - (e >> 6)) & (kHistSlots - 1)) * kHistStride + (e & 63) + i;
            const float d = dtab[di & 511];
            accL = tap(accL, d, lds[slot], fma);
            accR = tap(accR, d, lds[kHistSlots * kHistStride + slot], fma);
            e = (e + 128) & 1023;
            di += 64;
        }
        // audio_noasm.go:26-37 — second run
        e = (96 - v0 - pos) & 1023;
        di = d0 + 32 + i;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int slot = ((T - (e >> 6)) & (kHistSlots - 1)) * kHistStride + (e & 63) + i;
            const float d = dtab[di & 511];
            accL = tap(accL, d, lds[slot], fma);
            accR = tap(accR, d, lds[kHistSlots * kHistStride + slot], fma);
            e = (e + 128) & 1023;
            di += 64;
        }
        const float sL = accL / -1090519040.0f; // audio.go:390
        const float sR = accR / -1090519040.0f;
        const uint64_t fb = ((uint64_t)stream * a.n_frames + f) * 2304;
        const int o = t * 32 + i;
        switch (a.format) {
        case MPEGHIP_AUDIO_F32N: {
            float *out = reinterpret_cast<float *>(a.out) + fb + 2 * o;
            out[0] = sL;
            out[1] = sR;
            break;
        }
        case MPEGHIP_AUDIO_F32NLR: {
            float *out = reinterpret_cast<float *>(a.out) + fb;
            out[o] = sL;
            out[1152 + o] = sR;
            break;
        }
        case MPEGHIP_AUDIO_S16: { // audio.go:400-408
            int16_t *out = reinterpret_cast<int16_t *>(a.out) + fb + 2 * o;
            out[0] = (int16_t)(int32_t)(sL < 0 ? sL * 32768.0f : sL * 32767.0f);
            out[1] = (int16_t)(int32_t)(sR < 0 ? sR * 32768.0f : sR * 32767.0f);
            break;
        }
        default: { // MPEGHIP_AUDIO_F32, audio.go:409-417 (both constants are 2^31 in float32)
            float *out = reinterpret_cast<float *>(a.out) + fb + 2 * o;
            out[0] = sL * 2147483648.0f;
            out[1] = sR * 2147483648.0f;
            break;
        }
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
        ring[idx] = lds[(ch * kHistSlots + (T & (kHistSlots - 1))) * kHistStride + (e & 63)];
    }
}

MPG_HD void audio_store_vpos(const AudioArgs &a, uint32_t stream, int32_t vpos0)
{
    const int32_t Tend = kT0 + (int32_t)a.n_frames * 36;
    a.vpos_out[stream] = vpos_at(vpos0, Tend - 1);
}

} // namespace mpg
