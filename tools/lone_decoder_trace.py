#!/usr/bin/env python3
"""The lone decoder under rocprofv3 --kernel-trace --stats (GPU box): tests/golden/test.mpeg1video through mpeghost_video_decode, three
passes, with the host mirror (default) or without (argv[1] = 0) — which kernels a returned frame costs.
    rocprofv3 --kernel-trace --stats -d out -o trace -- python tools/lone_decoder_trace.py [0|1]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import hostlib  # noqa: E402

mirror = (sys.argv[1] if len(sys.argv) > 1 else "1") != "0"
dev = hostlib.host().mpeghost_device_create(0)
data = (ROOT / "tests" / "golden" / "test.mpeg1video").read_bytes()
frames = 0
for _ in range(3):
    dec = hostlib.HostVideo(data, device=dev)
    dec.set_host_mirror(mirror)
    while dec.decode() is not None:
        frames += 1
    dec.close()
print("host mirror %s: %d frames" % ("on" if mirror else "off", frames))
