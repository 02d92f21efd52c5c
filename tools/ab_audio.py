#!/usr/bin/env python3
"""A/B of the audio kernel's time slicing (MPEGHIP_AUDIO_CHUNKS) on BASELINE config 4 (256 streams x 100 frames)."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mpeg_amd import abi, desc, synth  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = abi.Context(0)
a = abi.AudioSynth(ctx, streams, desc.AUDIO_FMA_NONE)
smp = synth.audio_frames(streams, frames)
d_s, d_o = a.device_buffers(frames, desc.AUDIO_F32N)
a.upload(d_s, smp)
only = os.environ.get("MPEGHIP_AB_ONLY")
for chunks in ([only] if only else ["1", "2", "4", "8", "16", "auto"]):
    if chunks == "auto":
        os.environ.pop("MPEGHIP_AUDIO_CHUNKS", None)
    else:
        os.environ["MPEGHIP_AUDIO_CHUNKS"] = chunks
    for _ in range(2):
        a.synth_device(d_s, frames, desc.AUDIO_F32N, d_o)
    ctx.sync()
    ts = []
    for _ in range(5):
        ctx.timer_start()
        for _ in range(3):
            a.synth_device(d_s, frames, desc.AUDIO_F32N, d_o)
        ts.append(ctx.timer_stop_ms() / 3)
    ms = float(np.median(ts))
    nbytes = streams * frames * 18432
    print("chunks %-5s %8.3f ms  %7.1f G pairs/s  %7.1f GB/s alg (%.1f%% of 8 TB/s)" %
          (chunks, ms, streams * frames * 1152 / ms / 1e6, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80))
