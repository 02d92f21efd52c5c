#!/usr/bin/env python3
"""Where a lone Video's microseconds go (GPU box): the reference's BenchmarkDecodeVideo workload — tests/golden/test.mpeg1video (160x120)
and a written SIF / 1080p stream — through mpeghost_video_decode in a loop; parse / hand-over / read-back wall time per picture
(mpeghost_video_phase_seconds) beside the loop's own."""
import ctypes as C
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]


def main():
    H = C.CDLL(str(ROOT / "mpeg_amd" / "libmpeghost.so"))
    P = C.c_void_p
    H.mpeghost_device_create.restype, H.mpeghost_device_create.argtypes = P, [C.c_int]
    H.mpeghost_video_open.restype, H.mpeghost_video_open.argtypes = P, [P, C.c_char_p, C.c_size_t]
    H.mpeghost_video_decode.restype, H.mpeghost_video_decode.argtypes = C.c_int, [P, C.c_void_p]
    H.mpeghost_video_close.argtypes = [P]
    H.mpeghost_video_phase_seconds.argtypes = [P, C.POINTER(C.c_double * 3)]
    H.mpeghost_last_error.restype = C.c_char_p
    dev = H.mpeghost_device_create(0)
    if not dev:
        raise SystemExit(H.mpeghost_last_error().decode())
    frame = (C.c_uint8 * 256)()
    cases = [("test.mpeg1video 160x120", (ROOT / "tests" / "golden" / "test.mpeg1video").read_bytes())]
    import mpeg1_writer
    from mpeg_amd import synth
    for w, h in ((352, 240), (1920, 1080)):
        seq = synth.generate_sequence(w, h, 7, seed=0x5a, profile="natural")
        cases.append(("written %dx%d" % (w, h), mpeg1_writer.write_sequence(w, h, seq, repeat=4)))
    H.mpeghost_video_set_host_mirror.argtypes = [P, C.c_int]
    H.mpeghost_video_set_device_pack_from.argtypes = [P, C.c_uint32]
    modes = [("default", None, None), ("host-packed", None, 0), ("device-packed", None, 1), ("no mirror", 0, None)]
    for name, es in [(n + " [" + m[0] + "]", (e, m)) for n, e in cases for m in modes]:
        es, mode = es
        for rep in range(3):
            v = H.mpeghost_video_open(dev, es, len(es))
            if mode[1] is not None:
                H.mpeghost_video_set_host_mirror(v, mode[1])
            if mode[2] is not None:
                H.mpeghost_video_set_device_pack_from(v, mode[2])
            n, t0 = 0, time.perf_counter()
            while H.mpeghost_video_decode(v, frame) == 1:
                n += 1
            dt = time.perf_counter() - t0
            ph = (C.c_double * 3)()
            H.mpeghost_video_phase_seconds(v, C.byref(ph))
            H.mpeghost_video_close(v)
            print("%-42s run %d: %5d frames  %8.1f frames/s  %7.1f us/frame = parse %6.1f + submit %6.1f + read %6.1f + other %5.1f" %
                  (name, rep, n, n / dt, dt / n * 1e6, ph[0] / n * 1e6, ph[1] / n * 1e6, ph[2] / n * 1e6, (dt - ph[0] - ph[1] - ph[2]) / n * 1e6))


if __name__ == "__main__":
    main()
