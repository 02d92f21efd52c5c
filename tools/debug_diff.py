#!/usr/bin/env python3
"""Development aid: run a replicated typical batch on the GPU and report WHERE frames differ from the oracle."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mpeg_amd import abi, desc, synth
from oracle import pyoracle
w, h, streams, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctx = abi.Context(0)
seq = synth.generate_sequence(w, h, n, profile="typical")
store = abi.VideoStore(ctx, w, h, streams)
ref = pyoracle.OracleStore(w, h, 1, threads=1)
g = desc.geometry(w, h)
for i, s in enumerate(seq):
    b = store.upload(s.pics, s.mbs, s.coefs, replicate=streams)
    b.run()
    ctx.sync()
    ref.submit(s.pics, s.mbs, s.coefs)
    want = ref.read_planes(0, s.cur)
    bad_total = 0
    for st in range(streams):
        got = store.read_planes(st, s.cur)
        for pi, (a, bb) in enumerate(zip(want, got)):
            if not np.array_equal(a, bb):
                W = g["luma_w"] if pi == 0 else g["chroma_w"]
                d = np.nonzero(np.asarray(a) != np.asarray(bb))[0]
                ys, xs = d // W, d % W
                mb = 16 if pi == 0 else 8
                mbs = sorted(set(zip((ys // mb).tolist(), (xs // mb).tolist())))
                bad_total += len(d)
                if bad_total < 4000:
                    types = []
                    for (my, mx) in mbs[:6]:
                        k = [q for q in range(len(s.mbs)) if s.mbs["mb_x"][q] == mx and s.mbs["mb_y"][q] == my]
                        types.append((my, mx, int(s.mbs["flags"][k[0]]), int(s.mbs["cbp"][k[0]]), int(s.mbs["mv_x"][k[0]]), int(s.mbs["mv_y"][k[0]])) if k else (my, mx, "none"))
                    print("pic %d (type %d) stream %d plane %d: %d bytes differ in %d MBs; first (my,mx,flags,cbp,mvx,mvy): %s; rows in MB: %s" %
                          (i, s.picture_type, st, pi, len(d), len(mbs), types, sorted(set((ys % mb).tolist()))[:20]))
    print("picture %d: %d differing bytes over all streams" % (i, bad_total))
    if bad_total:
        break
