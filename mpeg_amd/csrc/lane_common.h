// lane_common.h — per-lane building blocks shared by every kernel of libmpeghip.
//
// The kernels are written as "lane functions": plain C++ that describes what ONE
// lane of a 64-wide wavefront does in one phase, with LDS passed in as a pointer.
// hipcc compiles them for gfx950 (the product).  The same source also builds with
// g++ into tests/kernel_emu (a test-only lane emulator that runs the 64 lanes of
// a wave in a loop) so the index arithmetic can be checked against the oracle on
// a machine without a GPU.  The emulator is NOT a fallback: libmpeghip never
// links it and refuses to create a context without a HIP device.
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MPG_HD __host__ __device__ __forceinline__
#define MPG_HDM __host__ __device__ __forceinline__ /* for static member functions */
#else
#define MPG_HD static inline __attribute__((always_inline))
#define MPG_HDM inline __attribute__((always_inline))
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define MPG_ON_DEVICE 1
#else
#define MPG_ON_DEVICE 0
#endif

// Emulator-only invariant checks (range proofs the device arithmetic relies on).
#if defined(MPG_EMU_CHECKS)
#include <assert.h>
#define MPG_CHECK(c) assert(c)
#else
#define MPG_CHECK(c) ((void)0)
#endif

namespace mpg {

// ---- unaligned little-endian loads / stores (gfx950 global memory handles any
// byte alignment in hardware; amdhsa enables unaligned-access-mode, so these
// become single global_load/store_dwordx2 instructions).
struct __attribute__((packed)) u64_unaligned { uint64_t v; };
struct __attribute__((packed)) u32_unaligned { uint32_t v; };

MPG_HD uint64_t ld64u(const uint8_t *p) { return reinterpret_cast<const u64_unaligned *>(p)->v; }
MPG_HD uint32_t ld32u(const uint8_t *p) { return reinterpret_cast<const u32_unaligned *>(p)->v; }
MPG_HD void st64u(uint8_t *p, uint64_t v) { reinterpret_cast<u64_unaligned *>(p)->v = v; }

// 16 bytes at any byte alignment (one global_load_dwordx4)
struct alignas(16) u8x16 { uint32_t v[4]; };
struct __attribute__((packed)) u8x16_unaligned { uint32_t v[4]; };
MPG_HD u8x16 ld128u(const uint8_t *p)
{
    const u8x16_unaligned *q = reinterpret_cast<const u8x16_unaligned *>(p);
    u8x16 r;
    r.v[0] = q->v[0];
    r.v[1] = q->v[1];
    r.v[2] = q->v[2];
    r.v[3] = q->v[3];
    return r;
}

// ---- naturally aligned 16-byte groups (one dwordx4 / ds_read_b128 each)
struct alignas(16) i16x8 { int16_t v[8]; };
struct alignas(16) i32x4 { int32_t v[4]; };
struct alignas(16) u32x4 { uint32_t v[4]; };
struct alignas(16) f32x4 { float v[4]; };

// ---- 24-bit multiply: v_mul_i32_i24 is full rate on CDNA, v_mul_lo_u32 is not.
// Callers guarantee both operands fit in 24 signed bits (checked in the emulator).
MPG_HD int32_t mul24(int32_t a, int32_t b)
{
    MPG_CHECK(a >= -(1 << 23) && a < (1 << 23) && b >= -(1 << 23) && b < (1 << 23));
#if MPG_ON_DEVICE
    return __mul24(a, b);
#else
    return a * b;
#endif
}

// The same for CHAINS of products (the dequantiser: level * scale * matrix entry): __mul24 is plain arithmetic to the
// optimiser, which reassociates the chain and, unable to prove 24 bits for its new operands, emits the quarter-rate
// v_mul_lo_u32.  The instruction itself keeps the association as written.
MPG_HD int32_t mul24_as_written(int32_t a, int32_t b)
{
    MPG_CHECK(a >= -(1 << 23) && a < (1 << 23) && b >= -(1 << 23) && b < (1 << 23));
#if MPG_ON_DEVICE
    int32_t r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a * b;
#endif
}

MPG_HD int32_t clampi(int32_t x, int32_t lo, int32_t hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ---- packed-byte averages on 4 pixels (video_noasm.go:14-26 roundAvg / bilinAvg)
// floor / ceil average per byte.  On the device this is V_LERP_U8:
//   D.byte[i] = (S0.byte[i] + S1.byte[i] + (S2.byte[i] & 1)) >> 1
MPG_HD uint32_t avg_floor_u8x4(uint32_t a, uint32_t b)
{
#if MPG_ON_DEVICE
    return __builtin_amdgcn_lerp(a, b, 0u);
#else
    return (a & b) + (((a ^ b) >> 1) & 0x7f7f7f7fu);
#endif
}

MPG_HD uint32_t avg_ceil_u8x4(uint32_t a, uint32_t b) // (a+b+1)>>1 per byte == roundAvg
{
#if MPG_ON_DEVICE
    return __builtin_amdgcn_lerp(a, b, 0x01010101u);
#else
    return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu);
#endif
}

// (a+b+c+d+2)>>2 per byte == bilinAvg, without widening:
//   p=(a+b)>>1, q=(c+d)>>1, e = both pair sums odd
//   (a+b+c+d+2)>>2 = (p+q+1+e)>>1 = ceil_avg(p,q) + (e & ~(p^q) & 1)
MPG_HD uint32_t avg4_u8x4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    uint32_t p = avg_floor_u8x4(a, b);
    uint32_t q = avg_floor_u8x4(c, d);
    uint32_t e = (a ^ b) & (c ^ d);
    uint32_t r = avg_ceil_u8x4(p, q);
    return r + (e & ~(p ^ q) & 0x01010101u);
}

MPG_HD uint64_t avg2_u8x8(uint64_t a, uint64_t b)
{
    uint32_t lo = avg_ceil_u8x4((uint32_t)a, (uint32_t)b);
    uint32_t hi = avg_ceil_u8x4((uint32_t)(a >> 32), (uint32_t)(b >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

MPG_HD uint64_t avg4_u8x8(uint64_t a, uint64_t b, uint64_t c, uint64_t d)
{
    uint32_t lo = avg4_u8x4((uint32_t)a, (uint32_t)b, (uint32_t)c, (uint32_t)d);
    uint32_t hi = avg4_u8x4((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), (uint32_t)(d >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// ---- write-back: 8 residuals + 8 prediction bytes -> 8 clamped output bytes
// (addBlockToDest / copyBlockToDest, video.go:943-971; clamp video.go:1014-1016).
// Device: per pixel pair v_cvt_pk_i16_i32 (saturating pack), v_perm_b32 (prediction
// bytes to 16-bit lanes), v_pk_add_i16 clamp, v_sat_pk_u8_i16 — 18 VALU for 8 pixels
// instead of ~40 scalar ones.  Saturating at +-32767 first is exact: anything that
// far out clamps to 0 / 255 either way.
MPG_HD uint32_t add_clamp_pack4(uint32_t pred4, int32_t v0, int32_t v1, int32_t v2, int32_t v3)
{
#if MPG_ON_DEVICE
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const s16x2 r01 = __builtin_amdgcn_cvt_pk_i16(v0, v1), r23 = __builtin_amdgcn_cvt_pk_i16(v2, v3);
    const uint32_t p01 = __builtin_amdgcn_perm(0u, pred4, 0x0c010c00u); // bytes 0,1 -> two zero-extended u16
    const uint32_t p23 = __builtin_amdgcn_perm(0u, pred4, 0x0c030c02u);
    const s16x2 s01 = __builtin_elementwise_add_sat(r01, __builtin_bit_cast(s16x2, p01));
    const s16x2 s23 = __builtin_elementwise_add_sat(r23, __builtin_bit_cast(s16x2, p23));
    uint32_t u01, u23;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(u01) : "v"(__builtin_bit_cast(uint32_t, s01)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(u23) : "v"(__builtin_bit_cast(uint32_t, s23)));
    return __builtin_amdgcn_perm(u23, u01, 0x05040100u); // {u01.b0, u01.b1, u23.b0, u23.b1}
#else
    const int32_t v[4] = {v0, v1, v2, v3};
    uint32_t out = 0;
    for (int c = 0; c < 4; c++) {
        int32_t r = v[c] < -32768 ? -32768 : (v[c] > 32767 ? 32767 : v[c]);
        int32_t x = (int32_t)((pred4 >> (8 * c)) & 0xff) + r;
        out |= (uint32_t)(x < 0 ? 0 : (x > 255 ? 255 : x)) << (8 * c);
    }
    return out;
#endif
}

MPG_HD uint64_t add_clamp_pack8(uint64_t pred, const int32_t (&v)[8])
{
    const uint32_t lo = add_clamp_pack4((uint32_t)pred, v[0], v[1], v[2], v[3]);
    const uint32_t hi = add_clamp_pack4((uint32_t)(pred >> 32), v[4], v[5], v[6], v[7]);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// bytes 1..4 of the 8-byte little-endian value {hi:lo} (one v_alignbyte_b32)
MPG_HD uint32_t shift_in_byte(uint32_t hi, uint32_t lo)
{
#if MPG_ON_DEVICE
    return __builtin_amdgcn_alignbyte(hi, lo, 1);
#else
    return (lo >> 8) | (hi << 24);
#endif
}

// Pin a wave-uniform value to a scalar register.  Besides documenting uniformity this stops
// the optimiser from folding scalar arithmetic back into per-lane selects (which turned
// s_mul_i32 into quarter-rate v_mul_lo_u32 in the address computation).
MPG_HD int32_t uniform(int32_t x)
{
#if MPG_ON_DEVICE
    return __builtin_amdgcn_readfirstlane(x);
#else
    return x;
#endif
}

// true if `pred` is false for every lane of the wavefront (wave-uniform branch key)
MPG_HD bool none_in_wave(bool pred)
{
#if MPG_ON_DEVICE
    return __builtin_amdgcn_ballot_w64(pred) == 0;
#else
    (void)pred;
    return false; // the emulator runs one lane at a time: never take the shortcut
#endif
}

// true if `pred` holds for every active lane of the wavefront.  The emulator runs one lane at a
// time and answers for that lane alone: only use it where both branches give the same result.
MPG_HD bool all_in_wave(bool pred)
{
#if MPG_ON_DEVICE
    return __builtin_amdgcn_ballot_w64(!pred) == 0;
#else
    return pred;
#endif
}

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4, no registers in
// between): lane l of the wave lands at lds_wave_base + 16*l.  The data is only guaranteed to be
// there after the next __syncthreads() (hipcc drains vmcnt in front of the barrier).
MPG_HD void copy16_to_lds(const void *g, void *lds_wave_base, int lane)
{
#if MPG_ON_DEVICE
    (void)lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
#else
    __builtin_memcpy(static_cast<char *>(lds_wave_base) + 16 * lane, g, 16);
#endif
}

// a store of something nobody on the device reads again (non-temporal: it does not take a line's place in L2)
template <typename T> MPG_HD void store_streaming(T *p, T v)
{
#if MPG_ON_DEVICE
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// ---- loads whose completion the KERNEL waits for (s_waitcnt vmcnt(N) placed by hand, wait_loads below), not the
// compiler: it forces vmcnt(0) in front of the first LDS access that follows a direct-to-LDS load, and it counts
// conservatively across branches; the reconstruction kernel wants "the two oldest loads are back" instead.
// They are inline assembly, so the compiler neither counts them nor waits for them.
//
// SIX direct-to-LDS loads (global_load_lds_dwordx4: 16 bytes per lane from global memory straight into LDS, no
// registers in between).  Load i: lane l fetches base[i] + off[i] (wave-uniform base, 32-bit per-lane byte offset)
// and lands at lds_wave_base + kAt_i + 16 l.  One asm statement: M0 (the LDS base) is written once, and the target
// offset of each load travels in the instruction's offset field, which the hardware adds to the LDS address AND to
// the global address (tools/microbench/lds_dma_probe4.hip) — the scalar base is moved back by the same amount.
// Loads complete in order, also mixed with register loads, and their data is in LDS as soon as s_waitcnt vmcnt lets
// the wave go (lds_dma_probe3.hip: 800 000 waves on cold lines).  (dwordx3 keeps the 16-byte lane stride and
// leaves holes: lds_dma_probe.hip.)
template <int kAt0, int kAt1, int kAt2, int kAt3, int kAt4>
MPG_HD void dma16x5_to_lds(const uint8_t *const (&g_base)[5], const uint32_t (&off)[5], void *lds_wave_base, int lane)
{
    static_assert(kAt0 >= 0 && kAt1 < 4096 && kAt2 < 4096 && kAt3 < 4096 && kAt4 < 4096, "13-bit signed offset field");
#if MPG_ON_DEVICE
    (void)lane;
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave_base); // LDS aperture: low 32 bits = offset
    asm volatile("s_mov_b32 m0, %10\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %5 offset:%11\n\t"
                 "global_load_lds_dwordx4 %1, %6 offset:%12\n\t"
                 "global_load_lds_dwordx4 %2, %7 offset:%13\n\t"
                 "global_load_lds_dwordx4 %3, %8 offset:%14\n\t"
                 "global_load_lds_dwordx4 %4, %9 offset:%15"
                 :
                 : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "s"(g_base[0] - kAt0), "s"(g_base[1] - kAt1),
                   "s"(g_base[2] - kAt2), "s"(g_base[3] - kAt3), "s"(g_base[4] - kAt4), "s"(base), "n"(kAt0), "n"(kAt1), "n"(kAt2),
                   "n"(kAt3), "n"(kAt4)
                 : "memory");
#else
    const int at[5] = {kAt0, kAt1, kAt2, kAt3, kAt4};
    for (int i = 0; i < 5; i++)
        __builtin_memcpy(static_cast<char *>(lds_wave_base) + at[i] + 16 * lane, g_base[i] + off[i], 16);
#endif
}
// The reconstruction kernel's five: one from the table array (its own base; lands at LDS offset kAtT = 0, so its offset field adds
// nothing) and four from ONE frame base.  That base sits kRcDmaBias below the stream's frames and every lane offset carries
// (kRcDmaBias - kAt_i) already (RcLane::cterm): the instruction's offset field, added to the LDS address AND to the global
// one, restores the difference — no scalar arithmetic per load.  off[i] INCLUDES that correction: lane l's piece i is at
// frame_base + off[i] + kAt_i.
#ifndef MPG_WINDOW_LOAD_BITS
#define MPG_WINDOW_LOAD_BITS "" // (cache-policy bits of the four window loads: sc0 / sc1 / nt measured, profiles/round6_e_*)
#endif
template <int kAtT, int kAt1, int kAt2, int kAt3, int kAt4>
MPG_HD void dma_table_and_windows(const uint8_t *table_base, const uint8_t *frame_base, const uint32_t (&off)[5], void *lds_wave_base, int lane)
{
    static_assert(kAtT == 0 && kAt1 > 0 && kAt1 < 4096 && kAt2 < 4096 && kAt3 < 4096 && kAt4 < 4096, "13-bit signed offset field");
#if MPG_ON_DEVICE
    (void)lane;
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave_base); // LDS aperture: low 32 bits = offset
    asm volatile("s_mov_b32 m0, %7\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %5\n\t"
                 "global_load_lds_dwordx4 %1, %6 offset:%8" MPG_WINDOW_LOAD_BITS "\n\t"
                 "global_load_lds_dwordx4 %2, %6 offset:%9" MPG_WINDOW_LOAD_BITS "\n\t"
                 "global_load_lds_dwordx4 %3, %6 offset:%10" MPG_WINDOW_LOAD_BITS "\n\t"
                 "global_load_lds_dwordx4 %4, %6 offset:%11" MPG_WINDOW_LOAD_BITS
                 :
                 : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "s"(table_base), "s"(frame_base), "s"(base), "n"(kAt1), "n"(kAt2),
                   "n"(kAt3), "n"(kAt4)
                 : "memory");
#else
    const int at[5] = {kAtT, kAt1, kAt2, kAt3, kAt4};
    __builtin_memcpy(static_cast<char *>(lds_wave_base) + at[0] + 16 * lane, table_base + off[0], 16);
    for (int i = 1; i < 5; i++)
        __builtin_memcpy(static_cast<char *>(lds_wave_base) + at[i] + 16 * lane, frame_base + off[i] + at[i], 16);
#endif
}
// ONE such load: lane l's 16 bytes from base + off (+ kAt) to LDS offset kAt + 16 l (recon_wide_kernel: a wave loads one window)
template <int kAt> MPG_HD void dma16_to_lds(const uint8_t *base, uint32_t off, void *lds_wave_base, int lane)
{
    static_assert(kAt >= 0 && kAt < 4096, "13-bit signed offset field");
#if MPG_ON_DEVICE
    (void)lane;
    const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" : : "v"(off), "s"(base), "s"(lbase), "n"(kAt) : "memory");
#else
    __builtin_memcpy(static_cast<char *>(lds_wave_base) + kAt + 16 * lane, base + off + kAt, 16);
#endif
}
// FOUR one-dword loads (global_load_lds_dword): lane l's dword i from base + off[i] (+ its offset field, as above) lands at LDS
// offset kAt + i * kStride + 4 l — the gather of a prediction window that leaves its plane (video_recon_lane.h), issued behind
// the wave's other direct-to-LDS loads: in flight together with them, no registers held, waited for by the same wait_loads.
template <int kAt, int kStride> MPG_HD void dma4x4_to_lds(const uint8_t *base, const uint32_t (&off)[4], void *lds_wave_base, int lane)
{
    static_assert(kAt >= 0 && kAt + 3 * kStride < 4096, "13-bit signed offset field");
#if MPG_ON_DEVICE
    (void)lane;
    const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dword %0, %4 offset:%6\n\t"
                 "global_load_lds_dword %1, %4 offset:%7\n\t"
                 "global_load_lds_dword %2, %4 offset:%8\n\t"
                 "global_load_lds_dword %3, %4 offset:%9"
                 :
                 : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(base), "s"(lbase), "n"(kAt), "n"(kAt + kStride), "n"(kAt + 2 * kStride),
                   "n"(kAt + 3 * kStride)
                 : "memory");
#else
    for (int i = 0; i < 4; i++)
        __builtin_memcpy(static_cast<char *>(lds_wave_base) + kAt + i * kStride + 4 * lane, base + off[i] + kAt + i * kStride, 4);
#endif
}
// one dword per lane into a register; only valid after wait_loads + settle() — and settle() it on EVERY path, used or
// not: until then the register belongs to the load, and the compiler must not hand it to something else
MPG_HD uint32_t load32_uncounted(const uint32_t *uniform_base, uint32_t byte_off)
{
#if MPG_ON_DEVICE
    uint32_t v;
    // (plain, not `nt`: read once, but a non-temporal load was 4 % slower on typical batches — profiles/r5_ab_*)
    asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(uniform_base) : "memory");
    return v;
#else
    return uniform_base[byte_off / 4];
#endif
}
// wait until at most `newer` of the loads above are still in flight (they complete in order)
template <int kNewer> MPG_HD void wait_loads()
{
#if MPG_ON_DEVICE
    static_assert(kNewer >= 0 && kNewer < 16, "vmcnt");
    __builtin_amdgcn_s_waitcnt(0x0f70 | kNewer); // gfx9 encoding: vmcnt[3:0], expcnt 7, lgkmcnt 15 = no wait on those
#endif
}
// ties a register loaded by load32_uncounted to the wait in front of it: its uses cannot move above this point
MPG_HD void settle(uint32_t &v)
{
#if MPG_ON_DEVICE
    asm volatile("" : "+v"(v)::"memory");
#else
    (void)v;
#endif
}

// Workgroup barrier that orders LDS only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also waits for every
// global store of the wave (vmcnt(0)) — on gfx950 stores count in vmcnt — which a kernel that streams its outputs
// out between barriers pays for at every one of them.  A wave whose direct-to-LDS loads the others are about to read
// waits for those itself first (wait_loads<0>).
MPG_HD void workgroup_barrier_lds()
{
#if MPG_ON_DEVICE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}

// the instruction scheduler moves nothing across this point (no instruction)
MPG_HD void sched_fence()
{
#if MPG_ON_DEVICE
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// hides a value's provenance from the optimiser (no instruction)
MPG_HD uint32_t opaque(uint32_t v)
{
#if MPG_ON_DEVICE
    asm("" : "+v"(v));
#endif
    return v;
}

// 16 bytes per lane to (wave-uniform base) + (32-bit lane offset): the scalar-base form of the store, so that no lane
// builds a 64-bit address (the compiler prefers v_lshl_add_u64 per lane when it sees base + offset itself)
// kStream: non-temporal — the bytes are not read again in this launch and should not push lines that are out of L2
#ifndef MPG_STREAM_STORE_BITS
#define MPG_STREAM_STORE_BITS "nt" // (with sc1 / sc0 sc1 — a wider scope, write-through — nothing changes: profiles/round6_e_*)
#endif
template <bool kStream> MPG_HD void store16_at(uint8_t *uniform_base, uint32_t off, const u32x4 &v)
{
#if MPG_ON_DEVICE
    typedef uint32_t vec4 __attribute__((ext_vector_type(4)));
    const vec4 d = {v.v[0], v.v[1], v.v[2], v.v[3]};
    if (kStream)
        asm volatile("global_store_dwordx4 %0, %1, %2 " MPG_STREAM_STORE_BITS : : "v"(off), "v"(d), "s"(uniform_base) : "memory");
    else
        asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(off), "v"(d), "s"(uniform_base) : "memory");
#else
    __builtin_memcpy(uniform_base + off, v.v, 16);
#endif
}

// one dword per lane to (wave-uniform base) + (32-bit lane offset) + (compile-time byte offset 0 .. 4095), non-temporal: the
// scalar-base form of the store — no lane builds a 64-bit address (left to itself the compiler keeps a 64-bit element index
// per lane: v_add, v_mov, v_lshl_add_u64 in front of every store of the audio kernel), and one scalar base serves a run of
// stores through the instruction's immediate offset
template <int kImm> MPG_HD void store32_streaming_at_imm(uint8_t *uniform_base, uint32_t off, float v)
{
    static_assert(kImm >= 0 && kImm < 4096, "global_store immediate offset");
#if MPG_ON_DEVICE
    asm volatile("global_store_dword %0, %1, %2 offset:%3 nt" : : "v"(off), "v"(v), "s"(uniform_base), "n"(kImm) : "memory");
#else
    __builtin_memcpy(uniform_base + off + kImm, &v, 4);
#endif
}

// XCD-aware block remap (MI355X: 8 XCDs, block b runs on XCD b%8, each XCD has its
// own L2).  Gives every XCD one contiguous range of work chunks so that
// neighbouring macroblocks — which share 128-byte destination lines and overlapping
// reference windows — meet in the same L2.  The grid is 8 * grid8 blocks (callers round up; surplus blocks find their index
// beyond the work and return): bijective on [0, 8 * grid8), and a multiply-add (any grid size cost a wave a dozen scalar
// instructions of quotient-and-remainder bookkeeping).
MPG_HD uint32_t xcd_chunk(uint32_t block, uint32_t grid8) { return (block & 7u) * grid8 + (block >> 3); }

} // namespace mpg
