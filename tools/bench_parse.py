#!/usr/bin/env python3
"""Host parser alone (no GPU needed): written 1080p streams (tests/mpeg1_writer.py) through mpeg::VideoBatch over a store that
swallows every request — pictures/s and ms per picture per parse thread, i.e. what the CPU side of the product costs.

Three streams: `escapes` (every coefficient an escape code: the writer of rounds 1-3, the parser's worst case), `table` (the
same typical-profile pictures with the run / level codes of Table B.5 wherever the table has one) and `natural` (levels as
encoders leave them — half of them +-1 — table-coded: what a real stream looks like).  `--root DIR` runs the parser of
ANOTHER source tree (a `git archive` of an older commit with its libraries built: tools/build_parse_history.sh) on the same
streams: the before / after of a parser change.

    python tools/bench_parse.py [--root DIR] [--threads 1,8,16] [--streams escapes,table,natural]"""
import argparse
import ctypes as C
import sys
import time
from pathlib import Path

ap = argparse.ArgumentParser()
ap.add_argument("--root", default=None)
ap.add_argument("--threads", default="1,8")
ap.add_argument("--streams", default="escapes,table,natural")
ap.add_argument("--pictures", type=int, default=7)
ap.add_argument("--repeat", type=int, default=1, help="the group of pictures this many times over (a longer stream of the same pictures)")
args = ap.parse_args()

HERE = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE / "tests"))
import mpeg1_writer  # noqa: E402  (always THIS tree's writer: the streams are the same for every parser)
from mpeg_amd import synth  # noqa: E402

ROOT = Path(args.root).resolve() if args.root else HERE
if args.root:   # the other tree's parser libraries through the other tree's loader
    for m in [k for k in sys.modules if k == "mpeg_amd" or k.startswith("mpeg_amd.")]:
        del sys.modules[m]
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
import hostlib  # noqa: E402

E = hostlib.host_emu()
E.host_emu_null_batch_store.restype = C.c_void_p
H = hostlib.host()
print("# parser of %s" % ROOT)
for kind in args.streams.split(","):
    profile = "natural" if kind == "natural" else "typical"
    seq = synth.generate_sequence(1920, 1080, args.pictures, seed=5, profile=profile)
    cache = Path("/tmp/bench_parse_%s_%d.es" % (kind, args.pictures))   # (the writer is a Python bit loop: a second per picture)
    if cache.exists():
        es = cache.read_bytes()
    else:
        es = mpeg1_writer.write_sequence(1920, 1080, seq, table=kind != "escapes")
        cache.write_bytes(es)
    if args.repeat > 1:   # header | group | end code -> header | group x repeat | end code (every group opens with its I picture)
        g0 = es.index(b"\x00\x00\x01\xb8")
        es = es[:g0] + es[g0:-4] * args.repeat + es[-4:]
    for threads in [int(x) for x in args.threads.split(",")]:
        streams = threads
        h = H.mpeghost_batch_open_store(E.host_emu_null_batch_store(), streams)
        if threads > 1:
            H.mpeghost_batch_set_threads(h, threads)
        keep = []
        for _ in range(streams):
            buf = C.create_string_buffer(es, len(es))
            keep.append(buf)
            H.mpeghost_batch_add_stream(h, buf, len(es))
        t0 = time.perf_counter()
        while H.mpeghost_batch_decode_all(h, 0) > 0:
            pass
        dt = time.perf_counter() - t0
        out = (C.c_uint64 * 2)()
        H.mpeghost_batch_counters(h, C.byref(out))
        H.mpeghost_batch_close(h)
        print("%-8s 1080p stream (%4.0f kB per picture), %2d stream(s) on %2d thread(s): %3d pictures in %7.1f ms = %6.0f pictures/s"
              " (%.3f ms per picture per thread)" % (kind, len(es) / len(seq) / args.repeat / 1e3, streams, threads, out[1], dt * 1e3, out[1] / dt,
                                                     dt * 1e3 * threads / out[1]), flush=True)
