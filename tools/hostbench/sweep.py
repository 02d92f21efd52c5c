#!/usr/bin/env python3
"""Staged hand-over rate against the number of putting threads (sparse and unit form), with the driver's phase split
(begin = waits for the staging buffer's previous use, puts, commit) on stderr.  python tools/hostbench/sweep.py"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from mpeg_amd import abi, synth  # noqa: E402
from mpeg_amd.shard import pin_to_node  # noqa: E402
from tools import hostbench  # noqa: E402

ctx = abi.Context(0)
print("numa node", ctx.numa_node(), "cpus bound", pin_to_node(ctx.numa_node()))
ctx.close()
seq = synth.generate_sequence(1920, 1080, 13, profile="typical")
for sparse in (True, False):
    for streams, threads in ((64, 1), (64, 8), (64, 16), (64, 32), (64, 64), (128, 64), (32, 32)):
        pps = hostbench.staged_submit_rate(0, 1920, 1080, seq, streams, threads, 1.0, verbose=True, sparse=sparse)
        print("%s  %3d pictures/call  %2d threads: %7.0f pictures/s  (%.3f ms per picture per thread)" % (
            "sparse" if sparse else "units ", streams, threads, pps, threads / pps * 1e3), flush=True)
