"""-m gpu: launches that mix what concurrent streams mix (mpeg_amd/mixed.py): stream s decodes the GOP of seed s % 16 at phase
s % 13 — I, P and B pictures of different streams in ONE launch, 208 distinct (content, phase) combinations, each against its
own oracle replay — and 25 / 50 / 75 % of the streams with the dense worst-case content beside typical ones, on both instances
of the reconstruction kernel and on the library's own choice; resident batches (host-packed at upload) and device-packed
staged commits of the same pictures."""
import numpy as np
import pytest

from mpeg_amd import abi, desc, mixed
from oracle import mixedcheck

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dense_share,tile", [(0.0, 0), (0.25, 0), (0.5, 1), (0.5, 2), (0.75, 0)])
def test_streams_at_different_gop_phases_with_different_content_in_one_launch(oracle, hip_ctx, dense_share, tile):
    w, h, n, gop = 352, 240, 208, 13
    wl = mixed.MixedWorkload(w, h, n, gop=gop, n_seeds=16, dense_share=dense_share, threads=8)
    assert len(wl.combos()) >= (208 if dense_share == 0 else 70)
    store = abi.VideoStore(hip_ctx, w, h, n)
    store.set_tile_policy(tile)
    try:
        batches = [store.upload(*wl.step_arrays(t)) for t in range(gop)]
        order = list(range(2 * gop + 3))
        for t in order:
            batches[t % gop].run()
        ok, text = mixedcheck.check(wl, store, order, threads=8)
        assert ok, text
        for b in batches:
            b.free()
    finally:
        store.close()


def test_mixed_pictures_through_device_packed_commits(oracle, hip_ctx):
    """The same mix handed over picture by picture through device-packed stages (pack_kernel sees I, P and B pictures, dense and
    typical ones, in one commit; the window of its waves is sized for the commit's largest picture)."""
    w, h, n, gop = 352, 240, 104, 13
    wl = mixed.MixedWorkload(w, h, n, gop=gop, n_seeds=8, dense_share=0.25, threads=8)
    store = abi.VideoStore(hip_ctx, w, h, n)
    try:
        order = list(range(gop + 4))
        for t in order:
            parts = []
            for s in range(n):
                x = wl.picture(s, t)
                pic = x.pics[0].copy()
                pic["stream"] = s
                parts.append((pic,) + desc.to_sparse(x.mbs, x.coefs))
            store.submit_staged_device(parts, threads=4, sync=False)
        store.sync()
        ok, text = mixedcheck.check(wl, store, order, threads=8)
        assert ok, text
    finally:
        store.close()
