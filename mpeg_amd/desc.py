"""Descriptor layouts of include/mpeghip.h as numpy dtypes, plus the frame-slot
rotation of the reference decoder (video.go:406-409, 430-433, 247-256)."""
from __future__ import annotations

import numpy as np

PIC_DTYPE = np.dtype([
    ("stream", "<u4"), ("cur", "u1"), ("fwd", "u1"), ("bwd", "u1"), ("flags", "u1"),
    ("mb_first", "<u4"), ("mb_count", "<u4"),
])
MB_DTYPE = np.dtype([
    ("pic", "<u4"), ("mb_x", "<u2"), ("mb_y", "<u2"), ("mv_x", "<i2"), ("mv_y", "<i2"),
    ("flags", "u1"), ("cbp", "u1"), ("qscale", "u1"), ("reserved0", "u1"),
    ("coef_off", "<u4"), ("reserved", "<u4", (3,)),
])
assert PIC_DTYPE.itemsize == 16 and MB_DTYPE.itemsize == 32

PIC_RGBA = 0x01
PIC_SPARSE = 0x02   # the picture's coefficient data is in the sparse hand-over form (to_sparse)
MB_INTRA, MB_REF_FWD, MB_REF_BWD, MB_COEF_RAW = 0x01, 0x02, 0x04, 0x08
COEF_UNIT = 128
SLOTS = 3

PIC_I, PIC_P, PIC_B = 1, 2, 3

AUDIO_F32N, AUDIO_F32NLR, AUDIO_F32, AUDIO_S16 = 0, 1, 2, 3
AUDIO_FMA_NONE, AUDIO_FMA_WINDOW = 0, 1
AUDIO_FRAME_INTS = 2 * 36 * 32

# video.go:1044-1053 (ISO 11172-2 zig-zag scan), natural index of scan position n
ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)


def geometry(width: int, height: int) -> dict:
    """video.go:314-322, 333-340."""
    mb_w, mb_h = (width + 15) >> 4, (height + 15) >> 4
    luma_w, luma_h = mb_w << 4, mb_h << 4
    luma = luma_w * luma_h
    chroma = luma // 4
    return dict(width=width, height=height, mb_w=mb_w, mb_h=mb_h, luma_w=luma_w, luma_h=luma_h,
                chroma_w=luma_w // 2, chroma_h=luma_h // 2, luma_bytes=luma, chroma_bytes=chroma,
                frame_bytes=luma + 2 * chroma + luma_w * 16, mb_count=mb_w * mb_h)


class SlotRotation:
    """Which of a stream's three frame slots is current / forward / backward.

    Mirrors decodePicture (video.go:406-409: `frameTemp = frameForward; if I/P
    { frameForward = frameBackward }`, and :430-433 `frameBackward =
    frameCurrent; frameCurrent = frameTemp` after a reference picture) and the
    output selection of Video.Decode (video.go:247-256)."""

    def __init__(self):
        self.cur, self.fwd, self.bwd = 0, 1, 2
        self.has_reference = False

    def begin(self, picture_type: int):
        """Slots to decode `picture_type` with: (cur, fwd, bwd)."""
        self._temp = self.fwd
        if picture_type in (PIC_I, PIC_P):
            self.fwd = self.bwd
        return self.cur, self.fwd, self.bwd

    def end(self, picture_type: int, no_delay: bool = False):
        """Finish the picture; returns the slot Video.Decode would hand out, or None."""
        if picture_type in (PIC_I, PIC_P):
            self.bwd = self.cur
            self.cur = self._temp
        if no_delay:
            return self.bwd
        if picture_type == PIC_B:
            return self.cur
        if self.has_reference:
            return self.fwd
        self.has_reference = True
        return None


def to_sparse(mbs, coefs, keep_zero_dc: bool = True):
    """The same picture in the SPARSE hand-over form (include/mpeghip.h: mpeghip_video_stage_put_sparse): per coded block
    a count word and one pair word `level << 16 | position << 2` per non-zero level of its unit (position order; an intra
    block's DC first, present even when 0), per snapshot block the count word 64 and its 64 int32 values.  Macroblocks name
    their words in order (every macroblock's coef_off = where the previous one's data ended).  -> (mbs with coef_off in
    dwords, words)"""
    mbs = np.array(mbs, dtype=MB_DTYPE, copy=True)
    n = len(mbs)
    raw_bytes = np.ascontiguousarray(coefs).view(np.uint8).reshape(-1)
    if n == 0:
        return mbs, np.zeros(0, np.uint32)
    n_units = raw_bytes.size // COEF_UNIT
    units16 = raw_bytes[:n_units * COEF_UNIT].view(np.int16).reshape(-1, 64)
    units32 = raw_bytes[:n_units * COEF_UNIT].view(np.int32).reshape(-1, 32)
    nb = _POPCOUNT6[mbs["cbp"].astype(np.int64) & 0x3f]
    intra = (mbs["flags"] & MB_INTRA) != 0
    raw = (mbs["flags"] & MB_COEF_RAW) != 0
    blk_mb = np.repeat(np.arange(n), nb)                                  # owning macroblock of every coded block
    first_blk = np.cumsum(nb) - nb
    blk_k = np.arange(len(blk_mb)) - first_blk[blk_mb]                    # ordinal among the macroblock's coded blocks
    b_raw, b_intra = raw[blk_mb], intra[blk_mb]
    unit = mbs["coef_off"].astype(np.int64)[blk_mb] + blk_k * np.where(b_raw, 2, 1)
    u = units16[np.where(b_raw, 0, unit)] if len(blk_mb) else np.zeros((0, 64), np.int16)
    mask = (u != 0) & ~b_raw[:, None]
    if keep_zero_dc:
        mask[:, 0] |= b_intra & ~b_raw
    cnt = mask.sum(axis=1)
    size = np.where(b_raw, 65, 1 + cnt)                                   # dwords per block
    blk_at = np.cumsum(size) - size
    mb_size = np.zeros(n, np.int64)
    np.add.at(mb_size, blk_mb, size)
    mbs["coef_off"] = np.cumsum(mb_size) - mb_size
    words = np.zeros(int(size.sum()), np.uint32)
    words[blk_at] = np.where(b_raw, 64, cnt)
    rows, cols = np.nonzero(mask)                                         # row-major: a block's positions ascending
    rank = np.arange(len(rows)) - (np.cumsum(cnt) - cnt)[rows]
    words[blk_at[rows] + 1 + rank] = (u[rows, cols].astype(np.uint16).astype(np.uint32) << 16) | (cols.astype(np.uint32) << 2)
    r = np.nonzero(b_raw)[0]
    if len(r):
        src = (unit[r][:, None] * 32 + np.arange(64)[None, :]).reshape(-1)
        dst = (blk_at[r][:, None] + 1 + np.arange(64)[None, :]).reshape(-1)
        words[dst] = units32.reshape(-1)[src].view(np.uint32)
    return mbs, words


_POPCOUNT6 = np.array([bin(i).count("1") for i in range(64)], np.int64)


def to_sparse_loop(mbs, coefs, keep_zero_dc: bool = True):
    """to_sparse written as the header describes it, block by block (tests compare the two)."""
    mbs = np.array(mbs, dtype=MB_DTYPE, copy=True)
    raw_bytes = np.ascontiguousarray(coefs).view(np.uint8).reshape(-1)
    units16 = raw_bytes.view(np.int16).reshape(-1, 64) if raw_bytes.size else np.zeros((0, 64), np.int16)
    units32 = raw_bytes.view(np.int32).reshape(-1, 32) if raw_bytes.size else np.zeros((0, 32), np.int32)
    out = []
    at = 0
    for k in range(len(mbs)):
        mb = mbs[k]
        nb = bin(int(mb["cbp"]) & 0x3f).count("1")
        unit = int(mb["coef_off"])
        mbs[k]["coef_off"] = at
        intra, raw = bool(mb["flags"] & MB_INTRA), bool(mb["flags"] & MB_COEF_RAW)
        for _ in range(nb):
            if raw:
                out.append(np.concatenate([[np.uint32(64)], units32[unit:unit + 2].reshape(-1).view(np.uint32)]).astype(np.uint32))
                unit += 2
                at += 65
                continue
            u = units16[unit]
            unit += 1
            pos = np.nonzero(u)[0]
            if intra and keep_zero_dc and (len(pos) == 0 or pos[0] != 0):
                pos = np.concatenate([[0], pos])
            pairs = (u[pos].astype(np.uint16).astype(np.uint32) << 16) | (pos.astype(np.uint32) << 2)
            out.append(np.concatenate([[np.uint32(len(pos))], pairs]).astype(np.uint32))
            at += 1 + len(pos)
    words = np.concatenate(out).astype(np.uint32) if out else np.zeros(0, np.uint32)
    return mbs, words
