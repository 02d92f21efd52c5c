#!/usr/bin/env python3
"""A/B of recon_kernel launch variants on resident batches (interleaved rounds, one process).
usage: ab_variants.py [streams] [variant ...]   variant = "mode,waves,blocks_per_cu" """
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mpeg_amd import abi, desc, synth  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = sys.argv[2:] or ["0,8,4", "1,8,4", "2,8,4", "2,4,8", "2,16,2", "1,4,8", "2,8,2", "2,8,3"]
ctx = abi.Context(0)
W, H = 1920, 1080
out = {}
for profile in ("typical", "dense"):
    seq = synth.generate_sequence(W, H, 5, profile=profile)
    if profile == "dense":
        seq = seq[:1] + [s for s in seq[1:] if s.picture_type == desc.PIC_P]
    store = abi.VideoStore(ctx, W, H, streams)
    batches = [store.upload(s.pics, s.mbs, s.coefs, replicate=streams) for s in seq]
    for b in batches:  # populate reference frames
        b.run()
    ctx.sync()
    timed = batches[1:]
    alg = sum(b.alg_bytes for b in timed)
    mbs = sum(b.n_mbs for b in timed)
    res = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            os.environ["MPEGHIP_RECON"] = v
            for b in timed:
                b.run()
            ctx.sync()
            ctx.timer_start()
            for _ in range(3):
                for b in timed:
                    b.run()
            ms = ctx.timer_stop_ms() / 3
            res[v].append(ms)
    print("== %s: %d streams, %d MB per pass, %.1f MB alg bytes" % (profile, streams, mbs, alg / 1e6))
    for v in variants:
        t = np.array(res[v])
        print("  variant %-8s median %8.3f ms  min %8.3f ms  -> %6.1f GB/s alg (%.1f%% of 8 TB/s), %.3f G MB/s" %
              (v, np.median(t), t.min(), alg / np.median(t) / 1e6, alg / np.median(t) / 1e6 / 80.0, mbs / np.median(t) / 1e6))
    for b in batches:
        b.free()
    store.close()
