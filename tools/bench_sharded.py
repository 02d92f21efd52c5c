#!/usr/bin/env python3
"""Throughput of mpeg::ShardedVideoBatch — the in-process multi-GPU driver of SURVEY §8(e): stream s -> device s mod G,
one VideoBatch + one host thread per device, no collective — end to end (bitstream parse on the host, sparse hand-over,
staged submits, reconstruction on the device; frames stay on the device).  Host-inclusive, NOT bench.py's `value`.

    python tools/bench_sharded.py --contexts 1,2,8 --streams-per-context 32 --threads 8 > profiles/rN_sharded.json

On a box with fewer GPUs than contexts the contexts share the devices (round robin): the figure is then the driver's own
cost — G host threads, G stores, G submits per tick — on ONE GPU, which is what can be measured without the 8-GPU node.
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contexts", default="1,2,8")
    ap.add_argument("--streams-per-context", type=int, default=32)
    ap.add_argument("--threads", type=int, default=8, help="parse threads per shard")
    ap.add_argument("--pictures", type=int, default=13)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    args = ap.parse_args()
    import hostlib
    import mpeg1_writer
    import torch
    from mpeg_amd import synth
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("needs a MI355X")
    seq = synth.generate_sequence(args.width, args.height, args.pictures, seed=0x5a)
    es = mpeg1_writer.write_sequence(args.width, args.height, seq)
    mb_per_pic = sum(len(s.mbs) for s in seq) / len(seq)
    H = hostlib.host()
    out = {"what": "mpeg::ShardedVideoBatch end to end: host parse -> sparse staged submits -> device, %dx%d written stream of %d pictures "
                   "(%d bytes), frames left on the device" % (args.width, args.height, args.pictures, len(es)),
           "gpus_visible": n_dev, "threads_per_shard": args.threads, "runs": []}
    for g in [int(x) for x in args.contexts.split(",")]:
        devices = []
        for k in range(g):
            d = H.mpeghost_device_create(k % n_dev)
            assert d, H.mpeghost_last_error()
            devices.append(d)
        n_streams = g * args.streams_per_context
        b = hostlib.HostSharded(n_streams, devices)
        b.set_threads(args.threads)
        for _ in range(n_streams):
            b.add_stream(es)
        t0, frames, ticks = time.perf_counter(), 0, 0
        while True:
            n = b.decode_all(fetch=False)
            if n == 0:
                break
            frames += n
            ticks += 1
        for d in devices:
            pass
        dt = time.perf_counter() - t0
        pictures = n_streams * args.pictures
        out["runs"].append({"contexts": g, "streams": n_streams, "frames_produced": frames, "ticks": ticks, "seconds": dt,
                            "pictures_per_s": pictures / dt, "macroblocks_per_s": pictures * mb_per_pic / dt,
                            "device_submits_per_shard": [b.counters(k)["device_submits"] for k in range(g)]})
        b.close()
        for d in devices:
            H.mpeghost_device_destroy(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
