"""TEST utility: write an MPEG-1 video elementary stream (ISO 11172-2) that carries a synthetic descriptor sequence
(mpeg_amd.synth.generate_sequence), so that full-size pictures can go through the product's PARSER and not only
through its descriptor ABI.  Not an encoder in the usual sense — there is no motion search, no transform, no rate
control: every macroblock of the sequence is written out with the type, vectors, quantiser and quantised levels
its descriptor holds.

Syntax choices that keep it small and exact:
  * one slice per macroblock row, address increment always 1 (no skipped macroblocks: a descriptor the generator
    calls "skipped" is written as a motion-compensated macroblock without pattern, which reconstructs the same);
  * every coded macroblock carries its quantiser (the *_quant types);
  * P and B macroblocks always carry their (single) vector: forward, or backward for descriptors that name the
    backward reference — the reference decoder never averages two predictions (video.go:626-630), and the
    generator's descriptors name exactly one;
  * coefficients are written with the run / level codes of Table B.5c-g wherever the table has a code for (run, |level|)
    and as escape codes (run 6 bits, level 8 / 16 bits) elsewhere — what an encoder does; `table=False` writes escapes only
    (always legal; the parser's worst case, and what rounds 1-3 measured);
  * the other tables are read from the host parser's own code list (mpeg_amd/host/iso11172_vlc_codes.h).
The tests decode the result with the product (parser -> descriptors -> device) AND with the oracle's parser and
compare both with the oracle's reconstruction of the ORIGINAL descriptors: a wrong code here cannot go unnoticed.
"""
from __future__ import annotations

import re
from pathlib import Path

import numpy as np

from mpeg_amd import desc

ROOT = Path(__file__).resolve().parent.parent
F_CODE = 3                      # vectors in [-64, 63] half-pels
FROM_COLMAJOR = np.array([(i % 8) * 8 + i // 8 for i in range(64)])   # natural (row-major) index -> column-major index

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7,
                   14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39,
                   46, 53, 60, 61, 54, 47, 55, 62, 63])   # scan position -> natural index (ISO 11172-2 2.4.3.7)


def _tables():
    text = (ROOT / "mpeg_amd" / "host" / "iso11172_vlc_codes.h").read_text()
    out = {}
    for m in re.finditer(r"mpg_vlc_code (\w+)\[\] = \{(.*?)\};", text, re.S):
        codes = {}
        for bits, value, dead in re.findall(r'\{"([01]+)", (-?(?:0x[0-9a-fA-F]+|\d+)), (\d)\}', m.group(2)):
            if dead == "0":
                codes.setdefault(int(value, 0), bits)
        out[m.group(1)] = codes
    return out


T = _tables()
MB_INTRA, MB_PATTERN, MB_BWD, MB_FWD, MB_QUANT = 1, 2, 4, 8, 16


class Bits:
    def __init__(self):
        self.chunks = []
        self.acc = 0
        self.n = 0

    def put(self, value: int, nbits: int):
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1))
        self.n += nbits
        if self.n >= 4096:
            self._spill()

    def code(self, bits: str):
        self.put(int(bits, 2), len(bits))

    def _spill(self):
        whole = self.n // 8
        rest = self.n - whole * 8
        self.chunks.append((self.acc >> rest).to_bytes(whole, "big"))
        self.acc &= (1 << rest) - 1
        self.n = rest

    def align(self):
        if self.n % 8:
            self.put(0, 8 - self.n % 8)

    def start_code(self, code: int):
        self.align()
        self.put(0x000001, 24)
        self.put(code, 8)

    def bytes(self) -> bytes:
        self.align()
        self._spill()
        return b"".join(self.chunks)


def _motion(b: Bits, target: int, pred: int) -> int:
    """One vector component (ISO 11172-2 2.4.4.2; decoded by video.go decodeMotionVector)."""
    f = 1 << (F_CODE - 1)
    lo, hi = -16 * f, 16 * f - 1
    assert lo <= target <= hi
    d = target - pred
    if d > hi:
        d -= 32 * f
    elif d < lo:
        d += 32 * f
    if d == 0:
        b.code(T["mpg_vlc_motion_code"][0])
        return target
    a = abs(d) - 1
    code, r = a // f + 1, a % f
    b.code(T["mpg_vlc_motion_code"][code if d > 0 else -code])
    b.put(r, F_CODE - 1)
    return target


def _dc(b: Bits, table: str, value: int, pred: int) -> int:
    d = value - pred
    size = abs(d).bit_length()
    b.code(T[table][size])
    if size:
        b.put(d if d > 0 else d + (1 << size) - 1, size)
    return value


COEFF = {v: bits for v, bits in T["mpg_vlc_dct_coeff"].items() if v not in (0x0001, 0xffff)}   # run << 8 | abs(level) -> code


def _coefficients(b: Bits, scan_levels, first: int, table: bool = True):
    """The block's (run, level) symbols for scan positions >= first, then end_of_block (ISO 11172-2 2.4.2.8, Table B.5c-g;
    read back by video.go:680-716 through the tree of video.go:1306-1419).  table: the code of Table B.5 where it has one for
    (run, |level|) — `1s` for (0, 1) as a non-intra block's very first coefficient, `11s` later — and the escape (run 6 bits,
    level 8 / 16 bits) elsewhere; False: escapes only (rounds 1-3's writer: the parser's worst case)."""
    prev = first - 1
    for pos in np.flatnonzero(scan_levels[first:]) + first:
        level = int(scan_levels[pos])
        assert -255 <= level <= 255
        run = int(pos) - prev - 1
        prev = int(pos)
        if table and run == 0 and abs(level) == 1:
            b.code("1" if (first == 0 and pos == 0) else "11")   # dct_coeff_first / dct_coeff_next
            b.put(1 if level < 0 else 0, 1)
            continue
        code = COEFF.get((run << 8) | abs(level)) if (table and run < 32 and abs(level) < 256) else None
        if code is not None:
            b.code(code)
            b.put(1 if level < 0 else 0, 1)
            continue
        b.put(0b000001, 6)
        b.put(run, 6)
        if -127 <= level <= 127:
            b.put(level, 8)
        elif level > 0:
            b.put(level, 16)              # 0000 0000 + 8 bits
        else:
            b.put(0x8000 | (level + 256), 16)
    b.put(0b10, 2)


def write_sequence(width: int, height: int, seq, frame_rate_code: int = 5, table: bool = True, repeat: int = 1, block_hook=None) -> bytes:
    """seq: list of mpeg_amd.synth.Submit (decode order, no MPEGHIP_MB_COEF_RAW macroblocks).  table: coefficients as Table B.5
    codes where the table has one (False: every coefficient as an escape code).  repeat: the group of pictures `repeat` times
    over (a longer stream of the same pictures for throughput runs: every group starts with its I picture).
    block_hook(picture index, macroblock index, block, intra) -> a string of '0' / '1' written IN THE PLACE of that block's
    coefficient symbols and end_of_block (None: the block as the descriptors have it) — for streams that are damaged on purpose."""
    g = desc.geometry(width, height)
    b = Bits()
    b.start_code(0xB3)
    b.put(width, 12)
    b.put(height, 12)
    b.put(1, 4)                 # pel aspect ratio 1.0
    b.put(frame_rate_code, 4)   # 5 = 30 pictures/s
    b.put(0x3FFFF, 18)          # variable bit rate
    b.put(1, 1)
    b.put(20, 10)               # vbv buffer size
    b.put(0, 1)                 # constrained parameters flag
    b.put(0, 1)                 # default intra matrix
    b.put(0, 1)                 # default non-intra matrix
    head = b.bytes()            # (start codes are byte aligned: the stream is header | group | group ... | end code)
    b = Bits()
    b.start_code(0xB8)          # group of pictures
    b.put(0, 25)
    b.put(1, 1)                 # closed gop
    b.put(0, 1)
    for pic_index, s in enumerate(seq):
        pt = s.picture_type
        b.start_code(0x00)
        b.put(0, 10)            # temporal reference (the reference decoder ignores it)
        b.put({desc.PIC_I: 1, desc.PIC_P: 2, desc.PIC_B: 3}[pt], 3)
        b.put(0xFFFF, 16)       # vbv delay
        if pt in (desc.PIC_P, desc.PIC_B):
            b.put(0, 1)
            b.put(F_CODE, 3)
        if pt == desc.PIC_B:
            b.put(0, 1)
            b.put(F_CODE, 3)
        b.put(0, 1)             # extra_bit_picture
        types = T["mpg_vlc_mb_type_i" if pt == desc.PIC_I else "mpg_vlc_mb_type_p" if pt == desc.PIC_P else "mpg_vlc_mb_type_b"]
        coefs = s.coefs.view(np.int16).reshape(-1, 64)
        mbs = s.mbs
        assert len(mbs) == g["mb_count"] and not (mbs["flags"] & desc.MB_COEF_RAW).any()
        for row in range(g["mb_h"]):
            b.start_code(row + 1)
            b.put(int(mbs[row * g["mb_w"]]["qscale"]), 5)
            b.put(0, 1)         # extra_bit_slice
            dc_pred = [128, 128, 128]
            fwd = [0, 0]
            bwd = [0, 0]
            for col in range(g["mb_w"]):
                m = mbs[row * g["mb_w"] + col]
                flags, cbp, unit = int(m["flags"]), int(m["cbp"]), int(m["coef_off"])
                intra = bool(flags & desc.MB_INTRA)
                b.code(T["mpg_vlc_mba_increment"][1])
                if intra:
                    assert cbp == 0x3f
                    b.code(types[MB_INTRA | MB_QUANT])
                    b.put(int(m["qscale"]), 5)
                    fwd, bwd = [0, 0], [0, 0]   # an intra macroblock resets the vector predictors
                else:
                    use_bwd = bool(flags & desc.MB_REF_BWD)
                    assert pt == desc.PIC_B or not use_bwd
                    t = (MB_BWD if use_bwd else MB_FWD) | ((MB_PATTERN | MB_QUANT) if cbp else 0)
                    b.code(types[t])
                    if cbp:
                        b.put(int(m["qscale"]), 5)
                    pred = bwd if use_bwd else fwd
                    pred[0] = _motion(b, int(m["mv_x"]), pred[0])
                    pred[1] = _motion(b, int(m["mv_y"]), pred[1])
                    if cbp:
                        b.code(T["mpg_vlc_coded_block_pattern"][cbp])
                    dc_pred = [128, 128, 128]   # ... and a non-intra one the DC predictors
                k = 0
                for blk in range(6):
                    if not cbp & (0x20 >> blk):
                        continue
                    scan = coefs[unit + k][FROM_COLMAJOR][ZIGZAG]
                    k += 1
                    damaged = block_hook(pic_index, row * g["mb_w"] + col, blk, intra) if block_hook else None
                    if damaged is not None and not intra:
                        b.code(damaged)
                        continue
                    if intra:
                        plane = 0 if blk < 4 else blk - 3
                        dc_table = "mpg_vlc_dct_dc_size_luma" if blk < 4 else "mpg_vlc_dct_dc_size_chroma"
                        dc_pred[plane] = _dc(b, dc_table, int(scan[0]), dc_pred[plane])
                        if damaged is not None:
                            b.code(damaged)
                            continue
                        _coefficients(b, scan, 1, table)
                    else:
                        assert scan.any()
                        _coefficients(b, scan, 0, table)
    group = b.bytes()
    b = Bits()
    b.start_code(0xB7)
    return head + group * repeat + b.bytes()
