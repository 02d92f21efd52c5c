"""TEST utility: write MPEG-1 Audio Layer II (MP2) elementary streams (ISO 11172-3) with chosen header modes and seeded random
content, so that the frame parser of the product (mpeg_amd/host/audio.cpp) and of the oracle are exercised beyond the one
golden file (mono, 64 kb/s, allocation table C): stereo, joint stereo with all four bounds, dual channel, mono; the four
allocation tables 3-B.2a-d (picked by bitrate x sampling rate x mode, audio.go:798-973); 32 / 44.1 / 48 kHz; with and without
CRC word; with and without padding slot.

Not an encoder: no filter bank, no psychoacoustics.  Every frame carries random bit allocations (thinned until the frame
fits), random scale factor selection information, random scale factors (63 among them: audio.go:452-453) and random sample
codes.  `expected_samples` restates, straight from ISO 11172-3 2.4.3.3 as the reference implements it (audio.go:440-490), what
the requantised sub-band samples of every written frame must be — the third leg beside the oracle's parser and the product's.
"""
from __future__ import annotations

import numpy as np

SAMPLERATE = [44100, 48000, 32000]
BITRATE = [32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384]
MODE_STEREO, MODE_JOINT, MODE_DUAL, MODE_MONO = 0, 1, 2, 3
SCALEFACTOR_BASE = [0x02000000, 0x01965FEA, 0x01428A30]
# ISO 11172-3 tables 3-B.2a-d as the reference folds them (audio.go:830-973): bitrate class, then sblimit / table, nbal and row
STEP1 = [[0, 0, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2], [0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 2]]
STEP2 = [[8, 8, 12], [27 | 64, 27 | 64, 27 | 64], [30 | 64, 27 | 64, 30 | 64]]
STEP3 = [[0x44, 0x44] + [0x34] * 10,
         [0x43] * 3 + [0x42] * 8 + [0x31] * 12 + [0x20] * 7]
STEP4 = [[0, 1, 2, 17], [0, 1, 2, 3, 4, 5, 6, 17], [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 17],
         [0, 1, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17], [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16],
         [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]]
QUANT = [(3, 1, 5), (5, 1, 7), (7, 0, 3), (9, 1, 10), (15, 0, 4), (31, 0, 5), (63, 0, 6), (127, 0, 7), (255, 0, 8), (511, 0, 9),
         (1023, 0, 10), (2047, 0, 11), (4095, 0, 12), (8191, 0, 13), (16383, 0, 14), (32767, 0, 15), (65535, 0, 16)]  # levels, grouped, bits


def table_of(mode: int, bitrate_index: int, samplerate_index: int):
    """-> (sblimit, which of the two nbal tables, its name A-D)"""
    t = STEP2[STEP1[0 if mode == MODE_MONO else 1][bitrate_index]][samplerate_index]
    sblimit, tab3 = t & 63, t >> 6
    return sblimit, tab3, {(27, 1): "A", (30, 1): "B", (8, 0): "C", (12, 0): "D"}[(sblimit, tab3)]


class _Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value: int, nbits: int):
        assert 0 <= value < (1 << nbits)
        self.v = (self.v << nbits) | value
        self.n += nbits

    def bytes(self, size: int) -> bytes:
        assert self.n <= size * 8, "frame overflow: %d bits in %d bytes" % (self.n, size)
        return (self.v << (size * 8 - self.n)).to_bytes(size, "big")


def write_stream(n_frames: int, mode: int, bitrate_index: int, samplerate_index: int, bound_code: int = 0, crc: bool = False,
                 seed: int = 1, fill: float = 0.85):
    """-> (stream bytes, frames): frames[i] = dict(allocation [2][32] -> quantiser index or 0, scfsi, scale_factor [2][32][3],
    codes [2][32][12][3], bound, sblimit): what was written, for expected_samples()."""
    rng = np.random.default_rng(seed)
    sblimit, tab3, _ = table_of(mode, bitrate_index, samplerate_index)
    channels = 1 if mode == MODE_MONO else 2
    bound = ((bound_code + 1) << 2) if mode == MODE_JOINT else (0 if mode == MODE_MONO else 32)
    bound = min(bound, sblimit)
    out, frames = [], []
    for f in range(n_frames):
        padding = int(samplerate_index == 0 and f % 2 == 1)      # 44.1 kHz: every other frame carries the padding slot
        size = 144000 * BITRATE[bitrate_index] // SAMPLERATE[samplerate_index] + padding
        budget = int((size * 8 - 32 - (16 if crc else 0)) * fill)
        # allocation indices: random, then thinned (highest subbands first) until the frame fits
        alloc = np.zeros((2, 32), np.int64)
        for sb in range(sblimit):
            nbal = STEP3[tab3][sb] >> 4
            for ch in range(channels if sb < bound else 1):
                alloc[ch, sb] = rng.integers(0, 1 << nbal)
            if sb >= bound:
                alloc[1, sb] = alloc[0, sb]
        scfsi = rng.integers(0, 4, size=(2, 32))

        def quant(ch, sb):
            a = int(alloc[ch, sb])
            return STEP4[STEP3[tab3][sb] & 15][a]

        def cost():
            bits = 0
            for sb in range(sblimit):
                nbal = STEP3[tab3][sb] >> 4
                bits += nbal * (2 if (sb < bound and channels == 2) else 1)
                for ch in range(channels):
                    q = quant(ch, sb)
                    if not q:
                        continue
                    bits += 2 + 6 * [3, 2, 1, 2][int(scfsi[ch, sb])]
                    if ch == 0 or sb < bound:                                    # joint part: one set of codes
                        levels, grouped, nb = QUANT[q - 1]
                        bits += 12 * (nb if grouped else 3 * nb)
            return bits

        sb = sblimit - 1
        while cost() > budget:
            alloc[:, sb] = 0
            sb = sb - 1 if sb > 0 else sblimit - 1
            if not alloc.any():
                break
        b = _Bits()
        b.put(0x7ff, 11)
        b.put(3, 2)
        b.put(2, 2)
        b.put(0 if crc else 1, 1)
        b.put(bitrate_index + 1, 4)
        b.put(samplerate_index, 2)
        b.put(padding, 1)
        b.put(int(rng.integers(0, 2)), 1)          # private bit
        b.put(mode, 2)
        b.put(bound_code if mode == MODE_JOINT else int(rng.integers(0, 4)), 2)   # mode extension (ignored outside joint stereo)
        b.put(int(rng.integers(0, 16)), 4)         # copyright, original, emphasis
        if crc:
            b.put(int(rng.integers(0, 65536)), 16)  # (the reference skips the CRC word: any value)
        for sb in range(sblimit):
            nbal = STEP3[tab3][sb] >> 4
            b.put(int(alloc[0, sb]), nbal)
            if sb < bound and channels == 2:
                b.put(int(alloc[1, sb]), nbal)
        for sb in range(sblimit):
            for ch in range(channels):
                if quant(ch, sb):
                    b.put(int(scfsi[ch, sb]), 2)
        sf = np.zeros((2, 32, 3), np.int64)
        for sb in range(sblimit):
            for ch in range(channels):
                if not quant(ch, sb):
                    continue
                vals = [int(x) for x in rng.integers(0, 64, size=3)]
                if rng.random() < 0.05:
                    vals[0] = 63
                k = int(scfsi[ch, sb])
                if k == 0:
                    s3 = vals
                elif k == 1:
                    s3 = [vals[0], vals[0], vals[2]]
                elif k == 2:
                    s3 = [vals[0]] * 3
                else:
                    s3 = [vals[0], vals[2], vals[2]]
                sf[ch, sb] = s3
                for v in ([s3[0], s3[1], s3[2]] if k == 0 else [s3[0], s3[2]] if k in (1, 3) else [s3[0]]):
                    b.put(v, 6)
        if mode == MODE_MONO:
            sf[1] = sf[0]
            scfsi[1] = scfsi[0]
        codes = np.zeros((2, 32, 12, 3), np.int64)
        for gr in range(12):
            for sb in range(sblimit):
                for ch in range(channels if sb < bound else 1):
                    q = quant(ch, sb)
                    if not q:
                        continue
                    levels, grouped, nb = QUANT[q - 1]
                    c = rng.integers(0, levels if grouped else (1 << nb), size=3)
                    codes[ch, sb, gr] = c
                    if grouped:
                        b.put(int(c[0] + c[1] * levels + c[2] * levels * levels), nb)
                    else:
                        for x in c:
                            b.put(int(x), nb)
                if sb >= bound:
                    codes[1, sb, gr] = codes[0, sb, gr]
        out.append(b.bytes(size))
        qidx = np.array([[quant(ch, sb) if sb < sblimit else 0 for sb in range(32)] for ch in range(2)])
        frames.append({"q": qidx, "scale_factor": sf, "codes": codes, "bound": bound, "sblimit": sblimit, "mode": mode})
    return b"".join(out), frames


def expected_samples(frame) -> np.ndarray:
    """int32 [2][36][32]: the requantised sub-band samples of one written frame in the layout the synthesis takes
    (include/mpeghip.h: sub-block t = (part * 4 + granule) * 3 + p), by the reference's arithmetic (audio.go:440-490) —
    including its simplification that the joint-stereo part of channel 1 is a copy of channel 0's SAMPLES (scaled with
    channel 0's scale factors, audio.go:407-412)."""
    out = np.zeros((2, 36, 32), np.int64)
    q, sfs, codes, bound, sblimit = frame["q"], frame["scale_factor"], frame["codes"], frame["bound"], frame["sblimit"]
    for gr in range(12):
        part = gr // 4
        for sb in range(sblimit):
            for ch in range(2):
                src = ch if (sb < bound and frame["mode"] != MODE_MONO) else 0
                if not q[src, sb]:
                    continue
                levels = QUANT[int(q[src, sb]) - 1][0]
                sf = int(sfs[src, sb, part])
                if sf == 63:
                    sf = 0
                else:
                    shift = sf // 3
                    sf = (SCALEFACTOR_BASE[sf % 3] + ((1 << shift) >> 1)) >> shift
                scale = 65536 // (levels + 1)
                adj = ((levels + 1) >> 1) - 1
                for p in range(3):
                    val = (adj - int(codes[src, sb, gr, p])) * scale
                    out[ch, gr * 3 + p, sb] = (val * (sf >> 12) + ((val * (sf & 4095) + 2048) >> 12)) >> 12
    return out.astype(np.int32)


# (name, mode, bitrate index, samplerate index, bound code, crc): every mode, every bound, every allocation table, every rate
CASES = [
    ("stereo_44k1_192_B", MODE_STEREO, 9, 0, 0, False),
    ("stereo_48k_128_A_crc", MODE_STEREO, 7, 1, 0, True),
    ("stereo_32k_96_D", MODE_STEREO, 5, 2, 0, False),
    ("stereo_44k1_64_C_crc", MODE_STEREO, 3, 0, 0, True),
    ("joint4_44k1_192_B", MODE_JOINT, 9, 0, 0, False),
    ("joint8_48k_160_A_crc", MODE_JOINT, 8, 1, 1, True),
    ("joint12_32k_64_D", MODE_JOINT, 3, 2, 2, False),
    ("joint16_44k1_96_C", MODE_JOINT, 5, 0, 3, False),      # bound 16 > sblimit 8: clamped (audio.go:285-287)
    ("joint16_32k_384_B_crc", MODE_JOINT, 13, 2, 3, True),
    ("dual_48k_256_A", MODE_DUAL, 11, 1, 0, False),
    ("dual_32k_320_B_crc", MODE_DUAL, 12, 2, 0, True),
    ("mono_44k1_64_A", MODE_MONO, 3, 0, 0, False),
    ("mono_32k_48_D_crc", MODE_MONO, 1, 2, 0, True),
    ("mono_48k_32_C", MODE_MONO, 0, 1, 0, False),
    ("mono_44k1_192_B", MODE_MONO, 9, 0, 0, False),
]
