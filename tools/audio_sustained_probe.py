#!/usr/bin/env python3
"""The 2048-stream audio launch repeated for seconds: does its duration drift (clocks / power state), and what does it settle at?
Prints the average per block of launches, and rocm-smi's clocks and power before, in the middle and after."""
import ctypes as C
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mpeg_amd import abi, desc, synth  # noqa: E402


def smi(tag):
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    keep = [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "junction", "edge"))]
    print("# rocm-smi %s: %s" % (tag, " | ".join(keep)[:600]), flush=True)


streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = abi.Context(0)
frames = 100
smp = synth.audio_frames(256, frames)
big = abi.AudioSynth(ctx, streams, desc.AUDIO_FMA_NONE)
d_s, d_o = big.device_buffers(frames, desc.AUDIO_F32N)
for t in range(streams // 256):
    big.upload(C.c_void_p(d_s.value + t * smp.nbytes), smp)
ctx.sync()
smi("before")
byts = streams * frames * 18432
t0 = time.perf_counter()
for b in range(blocks):
    ctx.timer_start()
    for _ in range(200):
        big.synth_device(d_s, frames, desc.AUDIO_F32N, d_o)
    ms = ctx.timer_stop_ms() / 200
    print("t = %5.2f s: %d streams, 200 launches, %.4f ms each, frac %.4f" % (time.perf_counter() - t0, streams, ms, byts / (ms * 1e-3) / 8e12), flush=True)
    if b == blocks // 2:
        smi("in the middle")
smi("after")
