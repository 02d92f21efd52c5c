#!/usr/bin/env python3
"""Host-inclusive rates (NOT bench.py's `value`): what one host thread + PCIe can feed.

 1. submit path: mpeghip_video_submit of one typical 1080p picture per call from pageable host arrays
    (validate + copy into pinned staging + H2D + kernel), back to back, 1 stream and 16 streams per call.
 2. BASELINE config 1: testdata/test.mpg (160x120, 278 pictures + 355 audio frames) through the product
    (host parser -> C ABI -> GPU, frames read back) vs the CPU oracle decoding the same file.
"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from mpeg_amd import abi, desc, synth  # noqa: E402


def submit_rate(ctx, streams, seconds=2.0):
    seq = synth.generate_sequence(1920, 1080, 13, profile="typical")
    store = abi.VideoStore(ctx, 1920, 1080, streams)
    batches = []
    for s in seq:
        pics = np.repeat(s.pics, streams)
        pics["stream"] = np.arange(streams)
        pics["mb_first"] = np.arange(streams) * len(s.mbs)
        mbs = np.tile(s.mbs, streams)
        mbs["pic"] = np.repeat(np.arange(streams), len(s.mbs))
        if streams > 1:  # every stream reads its own copy of the coefficients, as a real multi-stream submit would
            per = s.coefs.nbytes // desc.COEF_UNIT
            mbs["coef_off"] = mbs["coef_off"] + np.repeat(np.arange(streams, dtype=np.uint32) * per, len(s.mbs))
            coefs = np.tile(s.coefs, streams)
        else:
            coefs = s.coefs
        batches.append((pics, mbs, coefs))
    for b in batches:
        store.submit(*b)
    ctx.sync()
    t0, n, nbytes = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < seconds:
        for b in batches:
            store.submit(*b)
            n += streams
            nbytes += b[0].nbytes + b[1].nbytes + b[2].nbytes
    ctx.sync()
    dt = time.perf_counter() - t0
    store.close()
    return n / dt, nbytes / dt


def staged_rate(device, streams, threads, seconds=2.0):
    """mpeghip_video_stage_*: `streams` typical 1080p pictures per device call, each put (validation + packing into
    the device format, into pinned staging) from one of `threads` host threads — driven natively
    (tools/hostbench/staged_rate.cpp): a Python thread pool would be the bottleneck."""
    from tools import hostbench
    seq = synth.generate_sequence(1920, 1080, 13, profile="typical")
    pps = hostbench.staged_submit_rate(device, 1920, 1080, seq, streams, threads, seconds, verbose=True)
    mbs = [np.ascontiguousarray(s.mbs) for s in seq]
    coefs = [np.ascontiguousarray(s.coefs).view(np.uint8).reshape(-1) for s in seq]
    dense = float(np.mean([16 + m.nbytes + c.nbytes for m, c in zip(mbs, coefs)]))
    nz = [np.count_nonzero(c.view(np.int16).reshape(-1, 64), axis=1) for c in coefs]
    # the device format: 16-byte picture, 24 bytes per macroblock (chunk of 4 = 96), one word per block + one per non-zero level
    wire = float(np.mean([16 + len(m) * 24 + 4 * len(k) + 4 * np.where(k <= 32, k, 32).sum() for m, k in zip(mbs, nz)]))
    return pps, dense, wire


def main():
    import os
    ctx = abi.Context(0)
    for streams in (1, 16):
        pps, bps = submit_rate(ctx, streams)
        print("submit, %2d stream(s)/call: %8.0f pictures/s = %.3f G macroblocks/s, %.2f GB/s of descriptors+coefficients over PCIe"
              % (streams, pps, pps * 8160 / 1e9, bps / 1e9))
    ctx.close()

    import hostlib
    for streams, threads in ((64, 1), (64, 8), (64, 32), (256, 32), (512, 32)):
        threads = min(threads, os.cpu_count() or 1)
        pps, dense, wire = staged_rate(0, streams, threads)
        print("staged submit, %4d pictures/call put by %2d host thread(s): %8.0f pictures/s = %.3f G macroblocks/s; per picture "
              "%.2f MB as the ABI hands it over, %.2f MB in the device format (chunks + words) = %.1f GB/s over PCIe"
              % (streams, threads, pps, pps * 8160 / 1e9, dense / 1e6, wire / 1e6, pps * wire / 1e9))
    from oracle import pyoracle
    ps = (ROOT / "tests" / "golden" / "test.mpg").read_bytes()
    dev = hostlib.host().mpeghost_device_create(0)
    best = None
    for _ in range(5):
        m = hostlib.HostMpeg(ps, device=dev)
        m.set_enabled(True, False)
        t0, n = time.perf_counter(), 0
        while m.decode_video() is not None:
            n += 1
        dt = time.perf_counter() - t0
        m.close()
        best = dt if best is None or dt < best else best
    print("config 1, product on the GPU (host parse + per-picture submit + D2H of every frame): %d pictures in %.1f ms = %.0f pictures/s"
          % (n, best * 1e3, n / best))
    es = pyoracle.ps_extract(ps, 0xE0)[0]
    best = None
    for _ in range(5):
        d = pyoracle.VideoDecoder(es)
        t0, n = time.perf_counter(), 0
        while d.decode() is not None:
            n += 1
        dt = time.perf_counter() - t0
        d.close()
        best = dt if best is None or dt < best else best
    print("config 1, CPU oracle (C restatement of the reference, 1 thread): %d pictures in %.1f ms = %.0f pictures/s" % (n, best * 1e3, n / best))
    # the same file as N lockstep streams through mpeg::VideoBatch: one device call per tick
    es = pyoracle.ps_extract(ps, 0xE0)[0]
    import os
    many = min(32, os.cpu_count() or 1)
    for n_streams, fetch, threads in ((64, True, 1), (64, False, 1), (512, False, 1), (512, False, 8), (512, False, many),
                                      (2048, False, many)):
        b = hostlib.HostBatch(n_streams, device=dev, threads=threads)
        for _ in range(n_streams):
            b.add_stream(es)
        t0, frames = time.perf_counter(), 0
        while True:
            k = b.decode_all(fetch=fetch)
            if k == 0:
                break
            frames += k
        dt = time.perf_counter() - t0
        c = b.counters()
        b.close()
        print("config 1 x %d streams, VideoBatch, %d parse thread(s) (%s): %d pictures in %.1f ms = %.0f pictures/s, %d device calls for %d pictures"
              % (n_streams, threads, "frames read back" if fetch else "frames stay on the device", frames, dt * 1e3, frames / dt,
                 c["device_submits"], c["queued_pictures"]))
    # a 1080p stream (tests/mpeg1_writer.py over the bench's own descriptor generator) through the whole product
    # path — parse included — as N streams of mpeg::VideoBatch
    import mpeg1_writer
    seq = synth.generate_sequence(1920, 1080, 13, profile="typical")
    es = mpeg1_writer.write_sequence(1920, 1080, seq)
    print("1080p written stream: %d pictures, %.2f MB (%.0f kB per picture: every AC coefficient an escape code); decoded 4 times over"
          % (len(seq), len(es) / 1e6, len(es) / len(seq) / 1e3))
    end = es.rfind(b"\x00\x00\x01\xb7")
    es = (es[:end] if end >= 0 else es) * 4   # the same GOP four times (a new sequence header each time): longer runs
    for n_streams, threads in ((1, 1), (32, 1), (32, 8), (64, many), (128, many)):
        b = hostlib.HostBatch(n_streams, device=dev, threads=threads)
        for _ in range(n_streams):
            b.add_stream(es)
        t0, frames = time.perf_counter(), 0
        while True:
            k = b.decode_all(fetch=False)
            if k == 0:
                break
            frames += k
        dt = time.perf_counter() - t0
        c = b.counters()
        b.close()
        print("1080p stream x %d streams, VideoBatch, %d parse thread(s), frames stay on the device: %d pictures parsed + "
              "reconstructed in %.1f ms = %.0f pictures/s = %.3f G macroblocks/s (%d device calls)"
              % (n_streams, threads, c["queued_pictures"], dt * 1e3, c["queued_pictures"] / dt,
                 c["queued_pictures"] / dt * 8160 / 1e9, c["device_submits"]))
    hostlib.host().mpeghost_device_destroy(dev)


if __name__ == "__main__":
    main()
