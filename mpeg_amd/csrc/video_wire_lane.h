// video_wire_lane.h — coefficient units on the PCIe wire.
//
// The C ABI hands the library coefficient blocks as dense 128-byte units (64 int16 words; an int32 snapshot
// block is two units), and the reconstruction kernel reads them dense.  On the wire between the pinned
// staging buffer and the device they would be the bulk of a picture (≈ 235 of ≈ 283 bytes per typical
// macroblock) while holding ≈ 7 non-zero words of 64.  The staged submit (mpeghip_video_stage_*) therefore
// sends each unit as
//     header  uint32 : payload offset in dwords (24 bits) | count (8 bits)
//     payload         : count <= kWireMaxSparse: `count` entries  (word position 0..63) | (word << 16)
//                       count == kWireDense:     the 32 dwords of the unit as they are
// and wire_expand_kernel (mpeghip.hip) rebuilds the dense units in HBM before the reconstruction kernel runs:
// bit-identical to what a plain copy would have put there, so nothing downstream knows.  The host half
// (wire_pack_unit) runs inside mpeghip_video_stage_put, i.e. on the emitter's parser threads, in place of
// the memcpy it replaces.  Both halves live here so that tests/kernel_emu can check the round trip on the CPU.
#pragma once

#include "lane_common.h"

namespace mpg {

constexpr uint32_t kWireDense = 255;
constexpr uint32_t kWireMaxSparse = 31;     // 31 entries = 124 bytes: beyond that the unit travels as it is
constexpr uint32_t kWireUnitWords = 64;     // int16 words per unit
constexpr uint32_t kWireUnitDwords = 32;

// Which of the unit's 64 words are non-zero (bit k <=> word k).  Host only.
#if !MPG_ON_DEVICE && defined(__SSE2__)
} // namespace mpg
#include <emmintrin.h>
namespace mpg {
static inline uint64_t wire_nonzero_mask(const uint8_t *unit)
{
    const __m128i zero = _mm_setzero_si128();
    uint64_t mask = 0;
    for (int k = 0; k < 4; k++) { // 16 words per step
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(unit + k * 32));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(unit + k * 32 + 16));
        const __m128i z = _mm_packs_epi16(_mm_cmpeq_epi16(a, zero), _mm_cmpeq_epi16(b, zero)); // 0xff where zero
        mask |= (uint64_t)(uint16_t)~_mm_movemask_epi8(z) << (k * 16);
    }
    return mask;
}
#else
MPG_HD uint64_t wire_nonzero_mask(const uint8_t *unit)
{
    uint64_t mask = 0;
    for (uint32_t k = 0; k < kWireUnitWords; k++) {
        uint16_t w;
        memcpy(&w, unit + k * 2, 2);
        mask |= (uint64_t)(w != 0) << k;
    }
    return mask;
}
#endif

// Pack one unit at payload[used ...]; returns its header, advances `used` (dwords).  Host only.
MPG_HD uint32_t wire_pack_unit(const uint8_t *unit, uint32_t *payload, uint32_t &used)
{
    uint64_t mask = wire_nonzero_mask(unit);
    const uint32_t count = (uint32_t)__builtin_popcountll(mask);
    const uint32_t at = used;
    if (count > kWireMaxSparse) {
        memcpy(payload + used, unit, 128);
        used += kWireUnitDwords;
        return (at << 8) | kWireDense;
    }
    uint32_t *e = payload + used;
    while (mask) {
        const uint32_t pos = (uint32_t)__builtin_ctzll(mask);
        mask &= mask - 1;
        uint16_t w;
        memcpy(&w, unit + pos * 2, 2);
        *e++ = pos | ((uint32_t)w << 16);
    }
    used += count;
    return (at << 8) | count;
}

// Device half, one lane: lane (g, j) of a wave rebuilds 16-byte segment j of the wave's g-th unit.
//   phase A: zero the segment in the wave's LDS tile (8 units x 128 bytes)
//   phase B: (after a wave-level LDS hand-off) scatter entries j, j+8, j+16, j+24 of the unit
//   phase C: (after another hand-off) read the segment back and store it; a dense unit skips the tile
struct WireLane {
    uint32_t header;
    const uint32_t *payload; // of this unit's picture
    bool live;
};

MPG_HD void wire_phase_zero(uint8_t *tile, int lane)
{
    uint64_t *t = reinterpret_cast<uint64_t *>(tile + lane * 16);
    t[0] = 0;
    t[1] = 0;
}

MPG_HD void wire_phase_scatter(const WireLane &w, uint8_t *tile, int lane)
{
    const uint32_t count = w.header & 0xff;
    if (!w.live || count == kWireDense)
        return;
    const int g = lane >> 3, j = lane & 7;
    const uint32_t *e = w.payload + (w.header >> 8);
    uint16_t *t = reinterpret_cast<uint16_t *>(tile + g * 128);
    for (uint32_t k = (uint32_t)j; k < count; k += 8) {
        const uint32_t v = e[k];
        t[v & 63] = (uint16_t)(v >> 16);
    }
}

MPG_HD void wire_phase_store(const WireLane &w, const uint8_t *tile, int lane, uint8_t *dst_unit)
{
    if (!w.live)
        return;
    const int j = lane & 7;
    u32x4 v;
    if ((w.header & 0xff) == kWireDense) {
        const uint32_t *p = w.payload + (w.header >> 8) + j * 4;
        v = u32x4{{p[0], p[1], p[2], p[3]}};
    } else {
        v = *reinterpret_cast<const u32x4 *>(tile + lane * 16);
    }
    *reinterpret_cast<u32x4 *>(dst_unit + j * 16) = v;
}

} // namespace mpg
