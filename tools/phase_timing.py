#!/usr/bin/env python3
"""Where a wave of recon_kernel spends its wall time (development aid).  Needs the instrumented build
    hipcc <flags of mpeg_amd/_build.py> -DMPG_PHASE_TIMING mpeg_amd/csrc/mpeghip.hip -o mpeg_amd/libmpeghip_timing.so
(s_memtime stamps per phase, written by lane 0 of the first 60000 chunks; tools/gpu_phase_timing.sh builds it):
swaps it in for this process, runs pictures for 64 streams, prints per-phase medians."""
import ctypes as C
import shutil
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
lib = ROOT / "mpeg_amd" / "libmpeghip.so"
import os


def swap_in(src):  # via rename: the library may be mapped by this process, its inode must not be rewritten
    shutil.copy(src, str(lib) + ".tmp")
    os.replace(str(lib) + ".tmp", lib)


shutil.copy(lib, "/tmp/libmpeghip_keep.so")
swap_in(ROOT / "mpeg_amd" / "libmpeghip_timing.so")
try:
    from mpeg_amd import abi, desc, synth
    profile = sys.argv[1] if len(sys.argv) > 1 else "typical"
    streams = 64
    ctx = abi.Context(0)
    seq = synth.generate_sequence(1920, 1080, 4, profile=profile)
    store = abi.VideoStore(ctx, 1920, 1080, streams)
    batches = [store.upload(s.pics, s.mbs, s.coefs, replicate=streams) for s in seq]
    for b in batches:
        b.run()
    ctx.sync()
    for b in batches[1:]:   # a P and B pictures
        b.run()
    L = abi.load_library()
    L.mpeghip_debug_read_dump.restype = C.c_int
    L.mpeghip_debug_read_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    n = 60000
    buf = np.zeros((n, 8), np.uint64)
    L.mpeghip_debug_read_dump(store.h, buf.ctypes.data, buf.nbytes)
    t = buf[:, :7].astype(np.int64)
    ok = (t[:, 6] > t[:, 0]) & (t[:, 0] > 0)
    t, total = t[ok], buf[ok, 7]
    d = np.diff(t, axis=1)
    names = ["kernel arguments + chunk (scalar loads) arrived", "lane constants + every vector load issued",
             "residual pass 0 (waits for the entries)", "motion compensation (waits for the prediction loads)",
             "residual add (+ passes 1, 2)", "stores"]
    print("%s: %d waves sampled (last picture), s_memtime ticks (= shader clocks here: the median lifetime matches SQ_WAVE_CYCLES / SQ_WAVES)" % (profile, len(t)))
    life = t[:, 6] - t[:, 0]
    print("  wave lifetime            median %7.0f  mean %7.0f" % (np.median(life), life.mean()))
    for k, nme in enumerate(names):
        print("  %-52s median %7.0f  mean %7.0f  (%4.1f %%)" % (nme, np.median(d[:, k]), d[:, k].mean(), 100 * d[:, k].mean() / life.mean()))
    for nb in (0, 4, 8, 9, 12, 16, 24):
        m = total == nb
        if m.sum() > 50:
            print("  coded blocks = %2d: %6d waves, residual pass 0 mean %7.0f, lifetime mean %7.0f" % (nb, m.sum(), d[m, 2].mean(), life[m].mean()))
finally:
    swap_in("/tmp/libmpeghip_keep.so")
