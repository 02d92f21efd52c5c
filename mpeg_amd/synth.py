"""Synthetic descriptor batches.

No SIF / 1080p MPEG-1 asset exists and none can be made here, so throughput and
full-size parity runs use seeded synthetic macroblock-descriptor streams whose
statistics follow the reference's own test clip: macroblock type mix, coded
block pattern histogram, coefficients per block, half-pel mode mix.  Everything
is generated for ONE stream; independent streams reuse the descriptors (the
device replicates them, mpeghip_video_batch_upload_replicated).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import desc

VIDEO_SEED = 0x6D706567  # "mpeg"
AUDIO_SEED = 0x6D703200  # "mp2\0"

PREMULT = np.array([  # video.go:1077-1086
    32, 44, 42, 38, 32, 25, 17, 9, 44, 62, 58, 52, 44, 35, 24, 12,
    42, 58, 55, 49, 42, 33, 23, 12, 38, 52, 49, 44, 38, 30, 20, 10,
    32, 44, 42, 38, 32, 25, 17, 9, 25, 35, 33, 30, 25, 20, 14, 7,
    17, 24, 23, 20, 17, 14, 9, 5, 9, 12, 12, 10, 9, 7, 5, 2], dtype=np.int64)
INTRA_Q = np.array([  # video.go:1055-1064
    8, 16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37,
    19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
    22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58,
    26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83], dtype=np.int64)
NON_INTRA_Q = np.full(64, 16, dtype=np.int64)

# natural (row-major) index -> column-major position used by the coefficient stream
_NAT = np.arange(64)
TO_COLMAJOR = (_NAT % 8) * 8 + (_NAT // 8)


@dataclass
class Submit:
    """One picture of stream 0 in C-ABI form."""
    pics: np.ndarray
    mbs: np.ndarray
    coefs: np.ndarray       # uint8, multiple of 128 bytes
    picture_type: int
    cur: int
    fwd: int
    bwd: int
    out_slot: int | None    # slot Video.Decode would return after this picture


def mv_in_range(g: dict, mb_x, mb_y, mvx, mvy):
    """Vectorised statement of the reference's legal read range for copyMacroblock:
    every copyBlock read must stay inside [plane start, end of the frame buffer)
    (video_noasm.go:48-50 re-slices src to its capacity; outside it Go panics)."""
    mb_x, mb_y, mvx, mvy = (np.asarray(a, dtype=np.int64) for a in (mb_x, mb_y, mvx, mvy))
    lw, cw = g["luma_w"], g["chroma_w"]
    lsi = ((mb_y << 4) + (mvy >> 1)) * lw + (mb_x << 4) + (mvx >> 1)
    llast = lsi + (15 + (mvy & 1)) * lw + 15 + (mvx & 1)
    cmx = np.where(mvx < 0, -((-mvx) // 2), mvx // 2)  # truncation toward zero
    cmy = np.where(mvy < 0, -((-mvy) // 2), mvy // 2)
    csi = ((mb_y << 3) + (cmy >> 1)) * cw + (mb_x << 3) + (cmx >> 1)
    clast = csi + (7 + (cmy & 1)) * cw + 7 + (cmx & 1)
    cap_y = g["frame_bytes"]
    cap_cr = g["frame_bytes"] - g["luma_bytes"] - g["chroma_bytes"]
    return (lsi >= 0) & (llast < cap_y) & (csi >= 0) & (clast < cap_cr)


def dequant_premult(q, intra, qscale, qm):
    """numpy statement of video.go:719-744 (used only to fabricate in-contract RAW blocks)."""
    q = q.astype(np.int64)
    level = 2 * q
    if not intra:
        level = level + np.sign(q)
    level = (level * qscale * qm) >> 4
    even = (level & 1) == 0
    level = np.where(even, level - np.where(level > 0, 1, -1), level)
    level = np.clip(level, -2048, 2047)
    return np.where(q != 0, level * PREMULT, 0)


def _choose(rng, n, probs):
    return rng.choice(len(probs), size=n, p=np.asarray(probs) / np.sum(probs))


def _levels(rng, n_blocks, counts, first_pos, natural=False):
    """Quantised levels [n_blocks, 64] in NATURAL order: `counts[i]` coefficients at
    scan positions >= first_pos[i], biased toward the low-frequency end.  natural: magnitudes as encoders leave them (geometric:
    half of the levels +-1, a quarter +-2 ...; the "natural" profile) instead of uniform 1..8 with 2 % of large ones."""
    key = rng.random((n_blocks, 64)) + np.arange(64)[None, :] * 0.08
    key[np.arange(64)[None, :] < first_pos[:, None]] = np.inf
    rank = np.argsort(np.argsort(key, axis=1), axis=1)
    present = rank < counts[:, None]
    if natural:
        mag = np.minimum(rng.geometric(0.5, size=(n_blocks, 64)), 255)
    else:
        mag = rng.integers(1, 9, size=(n_blocks, 64))
        esc = rng.random((n_blocks, 64)) < 0.02
        mag = np.where(esc, rng.integers(1, 256, size=(n_blocks, 64)), mag)
    sign = np.where(rng.random((n_blocks, 64)) < 0.5, -1, 1)
    scan = np.where(present, mag * sign, 0)
    nat = np.zeros_like(scan)
    nat[:, desc.ZIGZAG] = scan  # scan position n lives at natural index ZIGZAG[n]
    return nat


def windows_inside(g, mb_x, mb_y, mvx, mvy):
    """True where the luma and chroma prediction windows of a macroblock lie inside their planes (rc_make_record's `inside`,
    video_recon_lane.h): the others are read the reference's way — linearly on, into the next row / plane / the pad
    (video_noasm.go:48-80) — which the kernels gather dword by dword (kRSlow)."""
    mvx, mvy = np.asarray(mvx, np.int64), np.asarray(mvy, np.int64)
    cmx, cmy = np.trunc(mvx / 2).astype(np.int64), np.trunc(mvy / 2).astype(np.int64)       # toward zero, video_noasm.go:35-36
    x0, y0 = mb_x * 16 + (mvx >> 1), mb_y * 16 + (mvy >> 1)
    cx0, cy0 = mb_x * 8 + (cmx >> 1), mb_y * 8 + (cmy >> 1)
    lw, lh = g["mb_w"] * 16, g["mb_h"] * 16
    return ((x0 >= 0) & (y0 >= 0) & (x0 + 16 + (mvx & 1) <= lw) & (y0 + 16 + (mvy & 1) <= lh) &
            (cx0 >= 0) & (cy0 >= 0) & (cx0 + 8 + (cmx & 1) <= lw // 2) & (cy0 + 8 + (cmy & 1) <= lh // 2))


def generate_picture(g: dict, picture_type: int, rng, profile: str = "typical", raw_fraction: float = 0.0,
                     prev_flags=None):
    """Macroblock descriptors + coefficient stream for one picture (pic index 0)."""
    n = g["mb_count"]
    mb_x = (np.arange(n) % g["mb_w"]).astype(np.int64)
    mb_y = (np.arange(n) // g["mb_w"]).astype(np.int64)
    mbs = np.zeros(n, desc.MB_DTYPE)
    mbs["mb_x"], mbs["mb_y"] = mb_x, mb_y
    qscale = rng.integers(1, 32, size=n)

    # "mc_copy" / "mc_horiz" / "mc_vert" / "mc_bilin": prediction only, every macroblock of a predicted picture with the
    # half-pel mode of the reference's BenchmarkCopyMacroblock{Copy,Horiz,Vert,Bilin} (video_test.go:105-118)
    # — the same vector for every macroblock, the reference's: (0,0), (1,0), (0,1), (3,3)
    mc_mode = {"mc_copy": (0, 0), "mc_horiz": (1, 0), "mc_vert": (0, 1), "mc_bilin": (3, 3)}.get(profile)
    if picture_type == desc.PIC_I:
        intra = np.ones(n, bool)
        skipped = np.zeros(n, bool)
    elif profile == "dense" or mc_mode:
        intra = np.zeros(n, bool)
        skipped = np.zeros(n, bool)
    else:
        kind = _choose(rng, n, [0.05, 0.09, 0.86])  # intra, skipped, inter
        intra, skipped = kind == 0, kind == 1

    # ---- motion
    if profile == "dense":
        mvx = rng.integers(-16, 16, size=n) * 2 + 1
        mvy = rng.integers(-16, 16, size=n) * 2 + 1
        bad = ~mv_in_range(g, mb_x, mb_y, mvx, mvy)
        mvx[bad], mvy[bad] = 1, 1
    elif mc_mode:
        mvx = np.full(n, mc_mode[0], np.int64)
        mvy = np.full(n, mc_mode[1], np.int64)
        bad = ~mv_in_range(g, mb_x, mb_y, mvx, mvy)
        mvx[bad], mvy[bad] = 0, 0
    else:
        mvx = rng.integers(-32, 33, size=n)
        mvy = rng.integers(-32, 33, size=n)
        bad = ~mv_in_range(g, mb_x, mb_y, mvx, mvy)
        mvx[bad], mvy[bad] = 0, 0
    ref_bwd = np.zeros(n, bool)
    if picture_type == desc.PIC_B:
        # 50 % bidirectional (backward prediction survives), 25 % fwd, 25 % bwd
        ref_bwd = rng.random(n) < 0.75
    if picture_type == desc.PIC_P:
        mvx[skipped], mvy[skipped] = 0, 0  # skipped P macroblocks: zero forward vector
    elif picture_type == desc.PIC_B:
        for i in np.nonzero(skipped)[0]:   # skipped B macroblocks repeat the previous vectors
            j = i - 1
            if j >= 0 and not intra[j] and mv_in_range(g, mb_x[i], mb_y[i], mvx[j], mvy[j]):
                mvx[i], mvy[i], ref_bwd[i] = mvx[j], mvy[j], ref_bwd[j]
            else:
                mvx[i], mvy[i] = 0, 0
    mvx[intra], mvy[intra] = 0, 0
    if profile == "typical_inside":  # diagnostic: no prediction window leaves its plane (what MPEG-1 allows an encoder: ISO 11172-2 2.4.4.2)
        mvx[~windows_inside(g, mb_x, mb_y, mvx, mvy)] = 0
        mvy[~windows_inside(g, mb_x, mb_y, mvx, mvy)] = 0
    if profile == "typical_fullpel":  # diagnostic: no half-pel interpolation anywhere
        mvx, mvy = mvx & ~1, mvy & ~1
        bad = ~mv_in_range(g, mb_x, mb_y, mvx, mvy)
        mvx[bad], mvy[bad] = 0, 0

    # ---- coded block pattern
    if profile == "dense":
        cbp = np.full(n, 0x3f)
    else:
        pop = _choose(rng, n, [0.30, 0.19, 0.20, 0.13, 0.11, 0.01, 0.06])
        order = np.argsort(rng.random((n, 6)), axis=1)
        bits = (np.argsort(order, axis=1) < pop[:, None])
        cbp = (bits * (0x20 >> np.arange(6))[None, :]).sum(axis=1)
    if profile in ("typical_nocoef", "typical_fullpel") or mc_mode:  # prediction only (intra stays coded)
        cbp = np.zeros(n, np.int64)
    cbp = np.where(intra, 0x3f, cbp)
    cbp = np.where(skipped, 0, cbp)

    raw = (rng.random(n) < raw_fraction) & (cbp != 0)
    flags = np.where(intra, desc.MB_INTRA, np.where(ref_bwd, desc.MB_REF_BWD, desc.MB_REF_FWD))
    flags = flags | np.where(raw, desc.MB_COEF_RAW, 0)

    nb = np.array([bin(int(c)).count("1") for c in range(64)])[cbp]
    units = nb * np.where(raw, 2, 1)
    coef_off = np.concatenate([[0], np.cumsum(units)[:-1]])
    total_units = int(units.sum())

    mbs["mv_x"], mbs["mv_y"] = mvx, mvy
    mbs["flags"], mbs["cbp"], mbs["qscale"] = flags, cbp, qscale
    mbs["coef_off"] = coef_off

    # ---- coefficient blocks
    coefs = np.zeros(max(total_units, 1) * desc.COEF_UNIT, np.uint8)
    if total_units:
        blk_mb = np.repeat(np.arange(n), nb)                      # owning macroblock of every coded block
        first = np.concatenate([[0], np.cumsum(nb)[:-1]])
        blk_k = np.arange(len(blk_mb)) - first[blk_mb]            # ordinal among the macroblock's coded blocks
        nblk = len(blk_mb)
        b_intra = intra[blk_mb]
        if profile == "dense":
            counts = np.full(nblk, 64)
        else:
            dc_only = rng.random(nblk) < 0.17
            counts = np.where(dc_only, 1, 1 + np.minimum(rng.geometric(1 / 7.7, size=nblk), 62))
        first_pos = np.where(b_intra, 1, 0)
        counts = np.where(b_intra, counts - 1, counts)            # intra: the DC is coded separately
        nat = _levels(rng, nblk, counts, first_pos, natural=profile == "natural")
        dc = rng.integers(16, 241, size=nblk)
        nat[:, 0] = np.where(b_intra, dc, nat[:, 0])
        b_raw = raw[blk_mb]
        off_bytes = (coef_off[blk_mb] + blk_k * np.where(b_raw, 2, 1)) * desc.COEF_UNIT

        qsel = ~b_raw
        if qsel.any():
            dst = coefs.view(np.int16)
            idx = (off_bytes[qsel] // 2)[:, None] + TO_COLMAJOR[None, :]
            dst[idx] = nat[qsel].astype(np.int16)
        if b_raw.any():
            r_idx = np.nonzero(b_raw)[0]
            vals = np.zeros((len(r_idx), 64), np.int64)
            for t, bi in enumerate(r_idx):
                m = blk_mb[bi]
                it = bool(intra[m])
                qm = INTRA_Q if it else NON_INTRA_Q
                v = dequant_premult(nat[bi], it, int(qscale[m]), qm)
                if it:
                    v[0] = int(nat[bi, 0]) * 256
                vals[t] = v
            dst = coefs.view(np.int32)
            idx = (off_bytes[b_raw] // 4)[:, None] + TO_COLMAJOR[None, :]
            dst[idx] = vals.astype(np.int32)
    return mbs, coefs


def gop_types(n_pictures: int):
    """Decode-order picture types: I P B B P B B ..."""
    out = [desc.PIC_I]
    while len(out) < n_pictures:
        out.extend([desc.PIC_P, desc.PIC_B, desc.PIC_B])
    return out[:n_pictures]


def generate_sequence(width: int, height: int, n_pictures: int, seed: int = VIDEO_SEED, profile: str = "typical",
                      raw_fraction: float = 0.0, rgba: bool = False, types=None):
    """A decode-order sequence of Submits for stream 0 with the reference's slot rotation."""
    g = desc.geometry(width, height)
    rng = np.random.default_rng(seed)
    rot = desc.SlotRotation()
    out = []
    for pt in (types or gop_types(n_pictures)):
        cur, fwd, bwd = rot.begin(pt)
        mbs, coefs = generate_picture(g, pt, rng, profile, raw_fraction)
        pics = np.zeros(1, desc.PIC_DTYPE)
        pics["stream"], pics["cur"], pics["fwd"], pics["bwd"] = 0, cur, fwd, bwd
        pics["flags"] = desc.PIC_RGBA if rgba else 0
        pics["mb_first"], pics["mb_count"] = 0, len(mbs)
        slot = rot.end(pt)
        out.append(Submit(pics, mbs, coefs, pt, cur, fwd, bwd, slot))
    return out


def alg_bytes(g: dict, sub: Submit) -> int:
    """32 + coefficient bytes + reference window + bytes written (+1024 RGBA) per macroblock."""
    m = sub.mbs
    intra = (m["flags"] & desc.MB_INTRA) != 0
    raw = (m["flags"] & desc.MB_COEF_RAW) != 0
    nb = np.array([bin(int(c)).count("1") for c in range(64)])[m["cbp"]]
    mvx, mvy = m["mv_x"].astype(np.int64), m["mv_y"].astype(np.int64)
    cmx = np.where(mvx < 0, -((-mvx) // 2), mvx // 2)
    cmy = np.where(mvy < 0, -((-mvy) // 2), mvy // 2)
    ref = (16 + (mvy & 1)) * (16 + (mvx & 1)) + 2 * (8 + (cmy & 1)) * (8 + (cmx & 1))
    per = 32 + nb * np.where(raw, 256, 128) + np.where(intra, 64 * nb, ref + 384)
    if sub.pics["flags"][0] & desc.PIC_RGBA:
        per = per + 1024
    return int(per.sum())


def audio_frames(n_streams: int, n_frames: int, seed: int = AUDIO_SEED, sblimit: int = 30):
    """int32 [n_streams, n_frames, 2, 36, 32]: bands < sblimit uniform in [-32768, 32767], rest zero."""
    rng = np.random.default_rng(seed)
    s = rng.integers(-32768, 32768, size=(n_streams, n_frames, 2, 36, 32), dtype=np.int32)
    s[..., sblimit:] = 0
    return s
