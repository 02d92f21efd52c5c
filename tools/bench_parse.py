#!/usr/bin/env python3
"""Host parser alone (no GPU needed): a written 1080p stream (tests/mpeg1_writer.py) through mpeg::VideoBatch over a
store that swallows every request — pictures/s per parse thread, i.e. what the CPU side of the product costs."""
import ctypes as C
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import hostlib  # noqa: E402
import mpeg1_writer  # noqa: E402
from mpeg_amd import synth  # noqa: E402

profile = sys.argv[1] if len(sys.argv) > 1 else "typical"
seq = synth.generate_sequence(1920, 1080, 7, seed=5, profile=profile)
es = mpeg1_writer.write_sequence(1920, 1080, seq)
E = hostlib.host_emu()
E.host_emu_null_batch_store.restype = C.c_void_p
H = hostlib.host()
for streams, threads in ((1, 1), (8, 1), (8, 8)):
    h = H.mpeghost_batch_open_store(E.host_emu_null_batch_store(), streams)
    if threads > 1:
        H.mpeghost_batch_set_threads(h, threads)
    keep = []
    for _ in range(streams):
        buf = C.create_string_buffer(es, len(es))
        keep.append(buf)
        H.mpeghost_batch_add_stream(h, buf, len(es))
    t0, n = time.perf_counter(), 0
    while H.mpeghost_batch_decode_all(h, 0) > 0:
        pass
    dt = time.perf_counter() - t0
    out = (C.c_uint64 * 2)()
    H.mpeghost_batch_counters(h, C.byref(out))
    H.mpeghost_batch_close(h)
    print("%s 1080p stream (%.0f kB per picture), %d stream(s), %d thread(s): %d pictures parsed in %.1f ms = %.0f pictures/s"
          " (%.2f ms per picture per thread)" % (profile, len(es) / len(seq) / 1e3, streams, threads, out[1], dt * 1e3, out[1] / dt,
                                                 dt * 1e3 * threads / out[1]))
