"""Synthetic batches in which every stream is somewhere else (bench.py's `mixed` leg, tests/test_gpu_mixed.py,
tools/sweep_dense_share.py).

The other benchmark batches replicate ONE stream's picture to every stream: all streams decode an I picture, or all a P, or
all a B picture in a launch, with the same words per chunk.  Real concurrent streams sit at different phases of their GOPs and
carry different content.  Here stream s decodes the GOP of seed `s % n_seeds`, `s % gop` pictures ahead of stream 0: one
launch reconstructs I, P and B pictures of different streams side by side, and `n_seeds x gop` distinct (content, phase)
combinations — each checked against its own CPU replay by the tests and bench.py (oracle/mixedcheck.py).  `dense_share` of the streams (s % dense_den < dense_den * share; dense_den = 4) decode the dense
worst-case profile instead (every block full, odd vectors), for the kernel-instance crossover (mpeghip.hip: kDenseBatchShare).

Batches are built in the unit form of the ABI (mpeghip_video_batch_upload), one picture per stream and step."""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import desc, synth


class MixedWorkload:
    def __init__(self, width: int, height: int, n_streams: int, gop: int = 13, n_seeds: int = 16, dense_share: float = 0.0,
                 rgba: bool = False, seed0: int = 0x4D58, threads: int = 8, dense_den: int = 4):
        self.w, self.h, self.n, self.gop, self.n_seeds = width, height, n_streams, gop, n_seeds
        self.dense_den = dense_den                                      # streams with s % dense_den < dense_quarters are dense
        self.dense_quarters = int(round(dense_share * dense_den))       # ("quarters": dense_den = 4 everywhere but in the fine sweep)
        kinds = [("typical", j) for j in range(n_seeds)]
        if self.dense_quarters:
            kinds += [("dense", j) for j in range(min(n_seeds, 2))]   # (dense pictures are 2.5 MB of units each: two seeds)
        with ThreadPoolExecutor(threads) as ex:                      # (numpy releases the GIL in the generator's sorts)
            seqs = list(ex.map(lambda k: synth.generate_sequence(width, height, gop, seed=seed0 + 97 * k[1] + (7919 if k[0] == "dense" else 0),
                                                                 profile=k[0], rgba=rgba), kinds))
        self.seqs = dict(zip(kinds, seqs))

    def combo(self, s: int):
        """(profile, seed, phase) of stream s."""
        if self.dense_quarters and s % self.dense_den < self.dense_quarters:
            return ("dense", (s // self.dense_den) % min(self.n_seeds, 2), s % self.gop)
        return ("typical", s % self.n_seeds, s % self.gop)

    def picture(self, s: int, t: int):
        """The Submit stream s decodes at step t."""
        profile, seed, phase = self.combo(s)
        return self.seqs[(profile, seed)][(t + phase) % self.gop]

    def combos(self):
        """distinct (profile, seed, phase) -> the first stream that has it"""
        out = {}
        for s in range(self.n):
            out.setdefault(self.combo(s), s)
        return out

    def step_arrays(self, t: int):
        """-> (pics, mbs, coefs) of step t: one picture per stream, unit form, ready for VideoStore.upload."""
        subs = [self.picture(s, t) for s in range(self.n)]
        n_mbs = np.array([len(x.mbs) for x in subs], np.int64)
        n_bytes = np.array([x.coefs.nbytes for x in subs], np.int64)
        mb_first = np.cumsum(n_mbs) - n_mbs
        unit_first = (np.cumsum(n_bytes) - n_bytes) // desc.COEF_UNIT
        pics = np.zeros(self.n, desc.PIC_DTYPE)
        mbs = np.empty(int(n_mbs.sum()), desc.MB_DTYPE)
        coefs = np.empty(int(n_bytes.sum()), np.uint8)
        at = 0
        for s, x in enumerate(subs):
            pics[s] = x.pics[0]
            m = mbs[mb_first[s]:mb_first[s] + n_mbs[s]]
            m[:] = x.mbs
            m["pic"] = s
            m["coef_off"] += np.uint32(unit_first[s])
            coefs[at:at + n_bytes[s]] = x.coefs.view(np.uint8).reshape(-1)
            at += int(n_bytes[s])
        pics["stream"] = np.arange(self.n)
        pics["mb_first"], pics["mb_count"] = mb_first, n_mbs
        return pics, mbs, coefs
