#!/usr/bin/env python3
"""bench.py — throughput of the MI355X reconstruction path.

A "step" is one pass of the hot path over one batch: ONE picture for every one of
`--streams` independent 1080p streams resident on this GPU (BASELINE.json config 5,
1024 streams per GPU; at N=1 that is the single-GPU shard).  Descriptor batches are
seeded synthetic ones already resident in HBM when the timed region
starts; steps walk a decode-order GOP (I P B B P B B ...) with the reference's frame
rotation.  With N GPUs every rank owns its own `--streams` streams (weak scaling, no
data-path collective — streams share nothing); `value` is the whole-job aggregate.

Prints ONE JSON line (rank 0): metric/value/... plus
  roofline     — algorithmic HBM bytes per launch / average launch time vs 8 TB/s
  cpu_baseline — the CPU oracle (restated reference algorithm) on a bounded sample
  audio        — the MP2 synthesis kernel on 256 stereo streams (BASELINE config 4)
  rgba_fused   — the same GOP with Frame.RGBA() of every picture fused into the kernel (BASELINE config 3's kernel)
  host_fed     — the same pictures handed over by host threads through the staged submit (PCIe inclusive; not `value`)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
MB_PER_1080P30_STREAM = 8160 * 30


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=26)
    ap.add_argument("--warmup", type=int, default=13)
    ap.add_argument("--streams", type=int, default=1024, help="independent 1080p streams per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--gop", type=int, default=13, help="pictures in the cycled decode-order GOP")
    ap.add_argument("--profile", default="typical", choices=["typical", "dense", "typical_nocoef", "typical_fullpel"],
                    help="typical / dense are the reported workloads; the other two are diagnostics (no residual / no half-pel)")
    ap.add_argument("--rgba", type=int, default=0, help="1: fuse Frame.RGBA into the reconstruction kernel")
    ap.add_argument("--rgba-streams", type=int, default=512,
                    help="streams of the secondary fused-RGBA leg (BASELINE config 3's kernel; 0 = skip; N=1 only)")
    ap.add_argument("--host-fed-seconds", type=float, default=0.0,
                    help="optional host-fed leg: pictures pushed through the staged submit from host threads for this many "
                         "seconds (N=1 only; off by default: it launches the reconstruction kernel on small batches, "
                         "which would blur a kernel trace of the run)")
    ap.add_argument("--audio-streams", type=int, default=256)
    ap.add_argument("--audio-frames", type=int, default=100)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--check", type=int, default=1, help="verify the final frames against the oracle (rank 0)")
    return ap.parse_args()


def cpu_baseline(args, seq, geom):
    """Time the oracle (CPU restatement of the reference's noasm algorithm) on a bounded
    sample of the same workload: T host threads, one independent stream each."""
    from oracle import pyoracle
    cores = os.cpu_count() or 1
    threads = max(1, cores)
    # single thread first: one stream, as many GOP pictures as fit in a quarter of the budget
    st1 = pyoracle.OracleStore(args.width, args.height, 1, threads=1)
    t0, n1, i = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < args.cpu_seconds * 0.25:
        s = seq[i % len(seq)]
        st1.submit(s.pics, s.mbs, s.coefs)
        n1 += len(s.mbs)
        i += 1
    r1 = n1 / (time.perf_counter() - t0)
    st1.close()
    # all cores: `threads` streams, the same picture each (independent frame stores)
    stT = pyoracle.OracleStore(args.width, args.height, threads, threads=threads)
    from mpeg_amd import desc
    t0, nT, i = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < args.cpu_seconds * 0.75:
        s = seq[i % len(seq)]
        pics = np.repeat(s.pics, threads)
        pics["stream"] = np.arange(threads)
        pics["mb_first"] = np.arange(threads) * len(s.mbs)
        mbs = np.tile(s.mbs, threads)
        mbs["pic"] = np.repeat(np.arange(threads), len(s.mbs))
        stT.submit(pics, mbs, s.coefs)
        nT += len(mbs)
        i += 1
    rT = nT / (time.perf_counter() - t0)
    stT.close()
    return {
        "value": rT, "unit": "macroblocks/s", "cores": threads, "kind": "port",
        "sample": "oracle (C restatement of the reference's pure-Go noasm path, gcc -O2): %d host threads x 1 "
                  "1080p stream each over the bench GOP (%s profile), %d macroblocks in %.1f s; single thread: %.3g "
                  "macroblocks/s" % (threads, args.profile, nT, args.cpu_seconds * 0.75, r1),
        "single_thread": r1,
    }


def main():
    args = parse_args()
    import torch

    from mpeg_amd.shard import Ranks
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    ranks = Ranks(backend="nccl", device_id=torch.device("cuda", local_rank))  # RCCL; only barrier + reductions
    world, rank, dist = ranks.world, ranks.rank, ranks.dist
    if args.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world), file=sys.stderr)

    from mpeg_amd import abi, desc, synth

    tstream = torch.cuda.Stream(device=local_rank)
    ctx = abi.Context(local_rank, tstream.cuda_stream)
    geom = desc.geometry(args.width, args.height)

    # ---- build the resident workload (per rank: its own streams, same seeded GOP)
    seq = synth.generate_sequence(args.width, args.height, args.gop, profile=args.profile, rgba=bool(args.rgba))
    if args.profile == "dense":
        # worst case per SURVEY §8(d): every timed picture is a dense P picture
        seq = seq[:1] + [s for s in seq[1:] if s.picture_type == desc.PIC_P]
    store = abi.VideoStore(ctx, args.width, args.height, args.streams)
    batches = [store.upload(s.pics, s.mbs, s.coefs, replicate=args.streams) for s in seq]
    gop_len = len(batches)
    ctx.sync()

    order = []

    def step(i):
        b = batches[i % len(batches)]
        b.run()
        order.append(i % len(batches))
        return b

    for i in range(args.warmup):
        step(i)
    acc = {"mbs": 0, "alg": 0, "ev_ms": 0.0}

    def timed_body():
        ctx.timer_start()
        for i in range(args.warmup, args.warmup + args.steps):
            b = step(i)
            acc["mbs"] += b.n_mbs
            acc["alg"] += b.alg_bytes
        acc["ev_ms"] = ctx.timer_stop_ms()  # HIP events on the stream the kernels run on

    # barrier + torch.cuda.synchronize() on both sides, MAX over ranks (mpeg_amd/shard.py)
    elapsed = ranks.timed(timed_body, device_sync=torch.cuda.synchronize)
    mbs_done, alg_done, ev_ms = acc["mbs"], acc["alg"], acc["ev_ms"]

    # ---- parity at full size (rank 0): all streams identical, and equal to the oracle's replay
    check = None
    if args.check and rank == 0:
        from oracle import pyoracle
        ref = pyoracle.OracleStore(args.width, args.height, 1, threads=1)
        for i in order:
            s = seq[i]
            ref.submit(s.pics, s.mbs, s.coefs)
        ok = True
        for slot in range(3):
            want = pyoracle.FNV_OFFSET
            for p in ref.read_planes(0, slot):
                want = pyoracle.fnv1a64(p, want)
            ok &= bool((store.hash_slots(slot) == np.uint64(want)).all())
        check = "bit-exact vs oracle on all %d streams x 3 slots after %d pictures" % (args.streams, len(order)) if ok else "MISMATCH"
        ref.close()
        if not ok:
            raise SystemExit("bench: frames differ from the oracle — result invalid")

    # ---- audio (BASELINE config 4), secondary metric
    audio = None
    if args.audio_streams > 0 and rank == 0:
        a = abi.AudioSynth(ctx, args.audio_streams, desc.AUDIO_FMA_NONE)
        smp = synth.audio_frames(args.audio_streams, args.audio_frames)
        d_s, d_o = a.device_buffers(args.audio_frames, desc.AUDIO_F32N)
        a.upload(d_s, smp)
        a.synth_device(d_s, args.audio_frames, desc.AUDIO_F32N, d_o)  # first launch from the zero state: checked below
        ctx.sync()
        aparity = None
        if args.check:
            from oracle import pyoracle
            got = a.download(d_o, args.audio_streams * args.audio_frames * 2304, desc.AUDIO_F32N).reshape(args.audio_streams, -1)
            probe = sorted({0, args.audio_streams // 2, args.audio_streams - 1})
            ok = True
            for st in probe:
                want = pyoracle.OracleSynth(1, 0).synth(smp[st:st + 1], desc.AUDIO_F32N).reshape(-1)
                ok &= bool(np.array_equal(want.view(np.uint32), got[st].view(np.uint32)))
            if not ok:
                raise SystemExit("bench: audio samples differ from the oracle — result invalid")
            aparity = "bit-exact vs oracle (no-FMA) on streams %s x %d frames" % (probe, args.audio_frames)
        for _ in range(2):
            a.synth_device(d_s, args.audio_frames, desc.AUDIO_F32N, d_o)
        ctx.sync()
        reps = 5
        ctx.timer_start()
        for _ in range(reps):
            a.synth_device(d_s, args.audio_frames, desc.AUDIO_F32N, d_o)
        ams = ctx.timer_stop_ms() / reps
        frames = args.audio_streams * args.audio_frames
        abytes = frames * 18432
        audio = {
            "metric": "MP2 stereo sample pairs/s", "value": frames * 1152 / (ams * 1e-3),
            "streams": args.audio_streams, "frames_per_launch": args.audio_frames, "ms_per_launch": ams,
            "realtime_streams_44k1": frames * 1152 / (ams * 1e-3) / 44100.0,
            "roofline": {"bound": "hbm", "achieved": abytes / (ams * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": abytes / (ams * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "audio_kernel<false, F32N> (DCT-32 + polyphase window, 4 waves per stream slice)"},
            "parity": aparity,
        }
        a.close()

    # ---- fused IDCT + MC + YCbCr->RGBA (BASELINE config 3's kernel), secondary: the same GOP with every picture
    # flagged MPEGHIP_PIC_RGBA, i.e. Frame.RGBA() of every decoded picture done inside the reconstruction kernel
    fused = None
    if args.rgba_streams > 0 and not args.rgba and rank == 0 and world == 1:
        for b in batches:
            b.free()
        batches = []
        n2 = args.rgba_streams
        store2 = abi.VideoStore(ctx, args.width, args.height, n2)
        seq2 = []
        for s_ in seq:
            pics = s_.pics.copy()
            pics["flags"] |= desc.PIC_RGBA
            seq2.append(pics)
        b2 = [store2.upload(p_, s_.mbs, s_.coefs, replicate=n2) for p_, s_ in zip(seq2, seq)]
        ctx.sync()
        order2 = []
        for i in range(args.warmup):
            b2[i % len(b2)].run()
            order2.append(i % len(b2))
        ctx.sync()
        mbs2 = alg2 = 0
        ctx.timer_start()
        for i in range(args.warmup, args.warmup + args.steps):
            b = b2[i % len(b2)]
            b.run()
            order2.append(i % len(b2))
            mbs2 += b.n_mbs
            alg2 += b.alg_bytes
        ms2 = ctx.timer_stop_ms()
        fparity = None
        if args.check:
            from oracle import pyoracle
            ref = pyoracle.OracleStore(args.width, args.height, 1, threads=1)
            for i in order2:
                ref.submit(seq2[i], seq[i].mbs, seq[i].coefs)
            ok = True
            for slot in range(3):
                want = ref.read_rgba(0, slot)
                for st in sorted({0, n2 - 1}):
                    ok &= bool(np.array_equal(np.asarray(store2.read_rgba(st, slot)).reshape(-1), want.reshape(-1)))
            ref.close()
            if not ok:
                raise SystemExit("bench: fused RGBA images differ from the oracle — result invalid")
            fparity = "RGBA images bit-exact vs oracle on streams %s x 3 slots after %d pictures" % (sorted({0, n2 - 1}), len(order2))
        fused = {
            "metric": "1080p macroblocks/sec, Frame.RGBA() of every picture fused into the reconstruction kernel",
            "value": mbs2 / (ms2 * 1e-3), "streams": n2, "steps": args.steps, "ms_per_step": ms2 / args.steps,
            "roofline": {"bound": "hbm", "achieved": alg2 / (ms2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "recon_wc_kernel<4, true> (the instance with Frame.RGBA fused; pictures flagged MPEGHIP_PIC_RGBA)"},
            "parity": fparity,
        }
        for b in b2:
            b.free()
        store2.close()

    # ---- host-fed rate (NOT `value`): the same pictures handed over by host threads through the staged submit,
    # i.e. validation + record expansion + wire packing on the host, PCIe, expansion + reconstruction on the device
    host_fed = None
    if args.host_fed_seconds > 0 and rank == 0 and world == 1:
        threads = min(32, os.cpu_count() or 1)
        pps = abi.staged_submit_rate(local_rank, args.width, args.height, seq, 64, threads, args.host_fed_seconds)
        mb_per_pic = float(np.mean([len(s.mbs) for s in seq]))
        host_fed = {"metric": "1080p macroblocks/sec handed over by host threads (mpeghip_video_stage_*), PCIe inclusive",
                    "value": pps * mb_per_pic, "pictures_per_s": pps, "host_threads": threads, "pictures_per_call": 64,
                    "realtime_1080p30_streams": pps * mb_per_pic / MB_PER_1080P30_STREAM}

    cpu = None
    if args.cpu_seconds > 0 and rank == 0 and world == 1:
        cpu = cpu_baseline(args, seq, geom)

    if rank == 0:
        total_mbs = mbs_done * world
        value = total_mbs / elapsed
        launch_ms = ev_ms / args.steps
        achieved = (alg_done / args.steps) / (launch_ms * 1e-3) / 1e9
        traffic = None
        tp = ROOT / "profiles" / "pmc_traffic.json"
        if tp.exists():
            try:  # measured for the default workloads only: same streams, same picture size
                t = json.loads(tp.read_text()).get(args.profile + ("_rgba" if args.rgba else ""), {})
                if t.get("streams") == args.streams and (args.width, args.height) == (1920, 1080):
                    traffic = t.get("hbm_bytes_per_launch")
                t2 = json.loads(tp.read_text()).get(args.profile + "_rgba", {})
                if fused and t2.get("streams") == args.rgba_streams and (args.width, args.height) == (1920, 1080):
                    fused["roofline"]["traffic"] = t2.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "1080p macroblocks/sec", "value": value, "unit": "macroblocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": "%d independent %dx%d MPEG-1 streams per GPU, one picture each per step, decode-order "
                                   "GOP of %d pictures (%s macroblock mix%s), descriptors resident in HBM" %
                                   (args.streams, args.width, args.height, gop_len, args.profile,
                                    ", fused RGBA" if args.rgba else ""),
                       "streams_per_gpu": args.streams, "macroblocks_per_step_per_gpu": mbs_done // args.steps,
                       "profile": args.profile, "rgba_fused": bool(args.rgba), "sharding": "by stream, no collective"},
            "realtime_1080p30_streams": value / MB_PER_1080P30_STREAM,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "recon_wc_kernel<4, %s> (one wave = 4 macroblocks, dense residual stage)" % ("true" if args.rgba else "false"), "alg_bytes_per_launch": alg_done // args.steps,
                         "avg_launch_ms": launch_ms},
            "cpu_baseline": cpu,
            "audio": audio,
            "rgba_fused": fused,
            "host_fed": host_fed,
            "parity": check,
        }
        print(json.dumps(line))
    for b in batches:
        b.free()
    store.close()
    ctx.close()
    ranks.close()


if __name__ == "__main__":
    main()
