"""TEST INFRASTRUCTURE (checker): the oracle's own decode of a mixed workload (mpeg_amd/mixed.py: every stream at its own GOP
phase with one of several contents) — one oracle stream per distinct (profile, seed, phase) combination — against a device
store, all streams x 3 slots.  Used by tests/test_gpu_mixed.py, tests/test_mixed_emu.py and bench.py's `mixed` leg; the
product never imports it."""
from __future__ import annotations

import numpy as np

from mpeg_amd import desc

from . import pyoracle


def oracle_replay(wl, order, threads: int = 16):
    """Every distinct combination's own decode of the steps in `order` on the oracle -> {combo: OracleStore stream index},
    store.  One oracle stream per combination, all of a step's pictures in one (threaded) submit."""
    combos = wl.combos()
    keys = list(combos)
    ref = pyoracle.OracleStore(wl.w, wl.h, len(keys), threads=threads)
    for t in order:
        subs = [wl.picture(combos[k], t) for k in keys]
        n_mbs = np.array([len(x.mbs) for x in subs], np.int64)
        n_bytes = np.array([x.coefs.nbytes for x in subs], np.int64)
        mb_first = np.cumsum(n_mbs) - n_mbs
        unit_first = (np.cumsum(n_bytes) - n_bytes) // desc.COEF_UNIT
        pics = np.concatenate([x.pics[:1] for x in subs])
        pics["stream"] = np.arange(len(keys))
        pics["mb_first"], pics["mb_count"] = mb_first, n_mbs
        mbs = np.concatenate([x.mbs for x in subs])
        mbs["pic"] = np.repeat(np.arange(len(keys)), n_mbs)
        mbs["coef_off"] += np.repeat(unit_first, n_mbs).astype(mbs["coef_off"].dtype)
        ref.submit(pics, mbs, np.concatenate([x.coefs.view(np.uint8).reshape(-1) for x in subs]))
    return {k: i for i, k in enumerate(keys)}, ref

def check(wl, store, order, rgba: bool = False, threads: int = 16):
    """All streams x 3 slots of `store` (abi.VideoStore) against each combination's oracle replay, by FNV-1a-64 of the
    planes (device-side hash per stream); with rgba also the RGBA images of one stream per combination.
    -> (ok, text)"""
    index, ref = oracle_replay(wl, order, threads)
    try:
        want = np.zeros((3, wl.n), np.uint64)
        per_combo = {}
        for k, i in index.items():
            hs = []
            for slot in range(3):
                hv = pyoracle.FNV_OFFSET
                for p in ref.read_planes(i, slot):
                    hv = pyoracle.fnv1a64(p, hv)
                hs.append(hv)
            per_combo[k] = hs
        for s in range(wl.n):
            for slot in range(3):
                want[slot, s] = per_combo[wl.combo(s)][slot]
        ok = all(bool((store.hash_slots(slot) == want[slot]).all()) for slot in range(3))
        text = "bit-exact vs oracle on all %d streams x 3 slots after %d steps: %d distinct (profile, seed, GOP phase) " \
               "combinations, each against its own oracle replay" % (wl.n, len(order), len(index))
        if ok and rgba:
            for k, s in wl.combos().items():
                for slot in range(3):
                    ok = ok and np.array_equal(np.asarray(store.read_rgba(s, slot)).reshape(-1), ref.read_rgba(index[k], slot).reshape(-1))
            text += "; RGBA images of one stream per combination"
        return ok, text
    finally:
        ref.close()
