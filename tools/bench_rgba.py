#!/usr/bin/env python3
"""Frame.RGBA on the device: the standalone conversion kernel over N resident 1080p frames, and the
reconstruction bench with MPEGHIP_PIC_RGBA (conversion of every written picture right after reconstruction)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mpeg_amd import abi, desc  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = abi.Context(0)
store = abi.VideoStore(ctx, 1920, 1080, streams)
g = desc.geometry(1920, 1080)
rng = np.random.default_rng(1)
y, cb, cr = (rng.integers(0, 256, n, dtype=np.uint8) for n in (store.info.luma_bytes, store.info.chroma_bytes, store.info.chroma_bytes))
store.write_planes(0, 0, y, cb, cr)
store.broadcast_slot(0, 0, 1, streams - 1)
for _ in range(2):
    store.rgba_convert(0, 0, streams)
ctx.sync()
ts = []
for _ in range(5):
    ctx.timer_start()
    for _ in range(3):
        store.rgba_convert(0, 0, streams)
    ts.append(ctx.timer_stop_ms() / 3)
ms = float(np.median(ts))
# algorithmic bytes per frame: visible Y + both chroma planes read once, width*height*4 written
rd = 1920 * 1080 + 2 * 960 * 540
wr = 1920 * 1080 * 4
gb = streams * (rd + wr) / ms / 1e6
print("rgba_kernel: %d frames of 1920x1080 in %.3f ms = %.0f frames/s, %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" %
      (streams, ms, streams / ms * 1e3, gb, gb / 80))
store.close()
ctx.close()
