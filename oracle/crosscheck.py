"""TEST INFRASTRUCTURE (checker only: imported by tests/ and by bench.py's parity step, never by the product).

Cross-stream reads at full size.  Benchmark batches replicate ONE stream's pictures to every stream of the store, so every
stream holds the same bytes: a per-stream base that is off by a stream (frames, words, tables, RGBA images) still reads
identical data and still hashes right.  This check makes such an error visible: a few far-apart streams get their OWN
reference content — seeded random planes written into all three slots, on the device and into one oracle store per
probed stream — then the non-intra pictures of the same replicated batches run once more, and each probed stream is
compared with its own oracle replay, plane by plane.  A stream that predicted from a neighbour's frames, or was written
into a neighbour's slots, differs.
"""
from __future__ import annotations

import numpy as np

from . import pyoracle


def probe_streams(n_streams: int):
    """first, second, one in the middle of the store (beyond the 4 GB mark at 1024 x 1080p), last"""
    return sorted({0, min(1, n_streams - 1), (517 * n_streams) // 1024, n_streams - 1})


def seeded_planes(geom, stream: int, slot: int):
    rng = np.random.default_rng(0x6D706567 + 977 * stream + slot)
    return (rng.integers(0, 256, geom["luma_bytes"], dtype=np.uint8), rng.integers(0, 256, geom["chroma_bytes"], dtype=np.uint8),
            rng.integers(0, 256, geom["chroma_bytes"], dtype=np.uint8))


def distinct_content_check(store, width, height, geom, n_streams, seq, batches, rgba=False, probes=None):
    """store: the device store (abi.VideoStore) the replicated `batches` (one per element of `seq`) belong to.  Runs the
    non-intra pictures of seq once (in order) on top of per-stream content; returns (ok, text)."""
    probes = probe_streams(n_streams) if probes is None else probes
    tail = [i for i, s in enumerate(seq) if not _all_intra(s)]
    if not tail:
        return True, "no predicted picture in the sequence: nothing to cross-check"
    refs = {}
    try:
        for st in probes:
            refs[st] = pyoracle.OracleStore(width, height, 1, threads=1)
            for slot in range(3):
                y, cb, cr = seeded_planes(geom, st, slot)
                refs[st].write_planes(0, slot, y, cb, cr)
                store.write_planes(st, slot, y, cb, cr)
        for i in tail:
            batches[i].run()
            for st in probes:
                refs[st].submit(seq[i].pics, seq[i].mbs, seq[i].coefs)
        bad = []
        for st in probes:
            for slot in range(3):
                for name, a, b in zip(("Y", "Cb", "Cr"), refs[st].read_planes(0, slot), store.read_planes(st, slot)):
                    if not np.array_equal(np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)):
                        bad.append("stream %d slot %d %s" % (st, slot, name))
            if rgba:
                cur = int(seq[tail[-1]].cur)
                if not np.array_equal(np.asarray(store.read_rgba(st, cur)).reshape(-1), refs[st].read_rgba(0, cur).reshape(-1)):
                    bad.append("stream %d slot %d RGBA" % (st, cur))
        text = "streams %s with their own reference content, %d predicted pictures on top: each bit-exact vs its own oracle replay" % (
            probes, len(tail))
        if bad:
            text = "CROSS-STREAM MISMATCH: " + ", ".join(bad[:8])
        return not bad, text
    finally:
        for r in refs.values():
            r.close()


def _all_intra(sub) -> bool:
    from mpeg_amd import desc
    return bool(((sub.mbs["flags"] & desc.MB_INTRA) != 0).all())
