"""mpeg_amd/mixed.py on the CPU (lane emulator): a launch with I, P and B pictures of different streams, typical and dense
content side by side, both kernel instances; every distinct combination against its own oracle replay."""
import numpy as np
import pytest

from mpeg_amd import mixed
from oracle import mixedcheck


class _EmuHashes:
    def __init__(self, oracle, store, n):
        self.oracle, self.store, self.n = oracle, store, n

    def hash_slots(self, slot):
        out = np.zeros(self.n, np.uint64)
        for s in range(self.n):
            hv = self.oracle.FNV_OFFSET
            for p in self.store.read_planes(s, slot):
                hv = self.oracle.fnv1a64(p, hv)
            out[s] = hv
        return out


@pytest.mark.parametrize("tile", [1, 2], ids=["int16", "int32"])
def test_mixed_launches_on_the_emulator(oracle, emu, tile):
    w, h, n, gop = 96, 64, 26, 13
    wl = mixed.MixedWorkload(w, h, n, gop=gop, n_seeds=3, dense_share=0.25, threads=2)
    kinds = {wl.picture(s, 0).picture_type for s in range(n)}
    assert len(kinds) == 3, "one launch must hold I, P and B pictures"
    emu.set_tile_policy(tile)
    try:
        st = emu.EmuStore(w, h, n)
        order = list(range(gop + 2))
        for t in order:
            st.submit(*wl.step_arrays(t))
        ok, text = mixedcheck.check(wl, _EmuHashes(oracle, st, n), order, threads=2)
        assert ok, text
    finally:
        emu.set_tile_policy(0)
