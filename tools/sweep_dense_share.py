#!/usr/bin/env python3
"""Where the two instances of the reconstruction kernel cross over (mpeghip.hip: kDenseBatchShare, kDenseWordsPerMb).

Launches in which a share of the STREAMS carries the dense worst-case content (every block full: dense units) and the rest
typical content, all at their own GOP phases (mpeg_amd/mixed.py), timed on the instance that transposes across lanes (policy 1, column
'int16') and on the instance for dense units (policy 2, column 'int32': the name of its first form), interleaved on one box (mpeghip_video_set_tile_policy).  Prints, per share: the batch's share of dense BLOCKS (what
launch_batch looks at), its sparse-form dwords per macroblock (what a device-packed commit looks at), ms per step on either
instance.   python tools/sweep_dense_share.py [streams [share of dense streams in per cent ...]] > profiles/round4_c_dense_share_crossover.txt"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mpeg_amd import abi, desc, mixed  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H, GOP, ROUNDS, REPS = 1920, 1080, 4, 3, 6
ctx = abi.Context(0)
print("# %d streams of %dx%d, GOP of %d (I P B B), stream s at phase s mod %d; a step = one picture per stream = one launch" % (streams, W, H, GOP, GOP))
print("# %-12s %-18s %-16s %-14s %-14s %s" % ("dense streams", "dense block share", "dwords per mb", "int16 ms/step", "int32 ms/step", "faster"))
for share in ([float(x) / 100 for x in sys.argv[2:]] or [0.0, 0.25, 0.5, 0.75, 1.0]):
    wl = mixed.MixedWorkload(W, H, streams, gop=GOP, n_seeds=4, dense_share=share, threads=8, dense_den=16)
    store = abi.VideoStore(ctx, W, H, streams)
    batches, blocks, dense_blocks, words, mbs = [], 0, 0, 0, 0
    for t in range(GOP):
        pics, m, c = wl.step_arrays(t)
        batches.append(store.upload(pics, m, c))
        # the batch's statistics as the packer counts them: coded blocks, blocks with more than 32 non-zero levels
        units = c.view(np.int16).reshape(-1, 64)
        nz = (units != 0).sum(axis=1)
        raw = np.repeat((m["flags"] & desc.MB_COEF_RAW) != 0, [bin(int(x)).count("1") for x in m["cbp"]]) if False else None
        blocks += len(nz)
        dense_blocks += int((nz > 32).sum())
        words += int(len(nz) + nz.sum())
        mbs += len(m)
    ctx.sync()
    ms = {1: [], 2: []}
    for r in range(ROUNDS):
        for policy in (1, 2):
            store.set_tile_policy(policy)
            for t in range(GOP):            # (warm: the other instance's code, the caches)
                batches[t].run()
            ctx.sync()
            ctx.timer_start()
            for _ in range(REPS):
                for t in range(GOP):
                    batches[t].run()
            ms[policy].append(ctx.timer_stop_ms() / (REPS * GOP))
    a, b = float(np.median(ms[1])), float(np.median(ms[2]))
    print("  %-12s %-18.3f %-16.1f %-14.4f %-14.4f %s (%+.1f %%)" % ("%.4g %%" % (share * 100), dense_blocks / blocks, words / mbs, a, b,
                                                                  "int16" if a < b else "int32", (max(a, b) / min(a, b) - 1) * 100), flush=True)
    for x in batches:
        x.free()
    store.close()
ctx.close()
