#!/usr/bin/env python3
"""Staged hand-over rate against the number of putting threads and the pictures per device call: DEVICE-packed stages (the puts
copy, pack_kernel validates and packs), the same with the pictures already in the pinned staging buffers (what is left when a
parser writes in place: commit + PCIe + device), and HOST-packed stages in the sparse and the unit form; the driver's phase
split (begin = waits for the staging buffer's previous use, puts, commit) on stderr.  python tools/hostbench/sweep.py [quick]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from mpeg_amd import abi, desc, synth  # noqa: E402
from mpeg_amd.shard import pin_to_node  # noqa: E402
from tools import hostbench  # noqa: E402

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
ctx = abi.Context(0)
print("numa node", ctx.numa_node(), "cpus bound", pin_to_node(ctx.numa_node()))
ctx.close()
seq = synth.generate_sequence(1920, 1080, 13, profile="typical")
wire = sum(32 * len(s.mbs) + 4 * len(desc.to_sparse(s.mbs, s.coefs)[1]) + 32 for s in seq) / len(seq)
print("a typical picture as the ABI's arrays (descriptors + sparse words): %.0f bytes" % wire)
secs = 0.6 if quick else 1.0
for mode, name in ((1, "device-packed, copied "), (2, "device-packed, in place"), (0, "host-packed, sparse   "), (-1, "host-packed, units    ")):
    grid = ((64, 1), (64, 2), (64, 4), (64, 8), (128, 8), (256, 8), (64, 16), (256, 16), (64, 32)) if mode > 0 else \
           ((64, 1), (64, 8), (64, 16), (64, 32), (128, 64))
    if quick:
        grid = grid[:6:2] if mode > 0 else grid[1:4:2]
    for streams, threads in grid:
        pps = hostbench.staged_submit_rate(0, 1920, 1080, seq[4:5] if mode == 2 else seq, streams, threads, secs, verbose=True,
                                           sparse=mode >= 0, device_pack=max(mode, 0))
        print("%s %3d pictures/call  %2d threads: %7.0f pictures/s  (%.3f ms per picture per thread; %.1f GB/s of ABI arrays)" % (
            name, streams, threads, pps, threads / pps * 1e3, pps * wire / 1e9), flush=True)
