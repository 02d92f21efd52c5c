#!/usr/bin/env python3
"""Is the 2048-stream audio launch slow because of its size in MEMORY (3.8 GB, beyond every cache) or because of its size as a
LAUNCH (10 240 workgroups in 8 generations)?  One launch of 2048 streams against 8 launches of 256 streams each, back to back on
one stream, every launch on its own buffers (the same 3.8 GB touched once)."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mpeg_amd import abi, desc, synth  # noqa: E402

ctx = abi.Context(0)
frames = 100
smp = synth.audio_frames(256, frames)
big = abi.AudioSynth(ctx, 2048, desc.AUDIO_FMA_NONE)
d_s, d_o = big.device_buffers(frames, desc.AUDIO_F32N)
for t in range(8):
    big.upload(C.c_void_p(d_s.value + t * smp.nbytes), smp)
parts = []
for t in range(8):
    a = abi.AudioSynth(ctx, 256, desc.AUDIO_FMA_NONE)
    ps, po = a.device_buffers(frames, desc.AUDIO_F32N)
    a.upload(ps, smp)
    parts.append((a, ps, po))
ctx.sync()
for rnd in range(3):
    for _ in range(2):
        big.synth_device(d_s, frames, desc.AUDIO_F32N, d_o)
    ctx.sync()
    ctx.timer_start()
    for _ in range(5):
        big.synth_device(d_s, frames, desc.AUDIO_F32N, d_o)
    one = ctx.timer_stop_ms() / 5
    for a, ps, po in parts:
        a.synth_device(ps, frames, desc.AUDIO_F32N, po)
    ctx.sync()
    ctx.timer_start()
    for _ in range(5):
        for a, ps, po in parts:
            a.synth_device(ps, frames, desc.AUDIO_F32N, po)
    eight = ctx.timer_stop_ms() / 5
    byts = 2048 * frames * 18432
    print("round %d: ONE launch of 2048 streams %.4f ms (frac %.4f)   EIGHT launches of 256 streams on their own buffers %.4f ms (frac %.4f)" % (
        rnd, one, byts / (one * 1e-3) / 8e12, eight, byts / (eight * 1e-3) / 8e12), flush=True)
