#!/usr/bin/env python3
"""bench.py — throughput of the MI355X reconstruction path.

A "step" is one pass of the hot path over one batch: ONE picture for every one of
`--streams` independent 1080p streams resident on this GPU (BASELINE.json config 5,
1024 streams per GPU; at N=1 that is the single-GPU shard).  Descriptor batches are
seeded synthetic ones already resident in HBM when the timed region
starts; steps walk a decode-order GOP (I P B B P B B ...) with the reference's frame
rotation.  With N GPUs every rank owns its own `--streams` streams (weak scaling, no
data-path collective — streams share nothing; the control plane is gloo: barrier + MAX
of the elapsed time); `value` is the whole-job aggregate.

Prints ONE JSON line (rank 0): metric/value/... plus
  roofline         — algorithmic HBM bytes per launch / average launch time (HIP events) vs 8 TB/s
  cpu_baseline     — the CPU oracle (restated reference algorithm) on a bounded sample
  dense            — the worst-case workload of SURVEY §8(d): dense P pictures (every block full, odd vectors)
  rgba_fused       — the typical GOP with Frame.RGBA() of every picture fused into the kernel (BASELINE config 3's kernel)
  dense_rgba_fused — the dense workload with Frame.RGBA() fused
  mixed            — every stream at its own GOP phase with one of 16 contents: I, P and B pictures of different streams in ONE launch
  audio            — the MP2 synthesis kernel on 256 stereo streams (BASELINE config 4)
  host_fed         — (optional) the same pictures handed over by host threads through the staged submit (PCIe inclusive)
Every video leg carries its own roofline object and parity string (all streams x 3 slots against the oracle's replay).
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
KERNEL = {False: "recon_kernel<1, false, T> (a wave reconstructs one chunk of 4 macroblocks; sparse coefficient entries, prediction windows "
                 "by direct-to-LDS loads, records that carry their own address arithmetic; int16 coefficient tile, 8 waves per SIMD; T = true: the "
                 "IDCT's transposition across lanes (typical batches), false: through LDS + the short dequantisation of dense units (batches "
                 "with dense units)); launches that leave the wave slots empty (one or two pictures) run "
                 "recon_wide_kernel: four waves per chunk",
          True: "recon_kernel<1, true, T> / recon_wide_kernel<true> (the instances with Frame.RGBA fused; pictures flagged MPEGHIP_PIC_RGBA)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=26)
    ap.add_argument("--warmup", type=int, default=13)
    ap.add_argument("--streams", type=int, default=1024, help="independent 1080p streams per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--gop", type=int, default=13, help="pictures in the cycled decode-order GOP")
    ap.add_argument("--profile", default="typical", choices=["typical", "dense", "typical_nocoef", "typical_fullpel", "typical_inside", "mc_copy", "mc_horiz", "mc_vert", "mc_bilin"],
                    help="workload of the PRIMARY leg (`value`): typical is the reported one; the others are diagnostics")
    ap.add_argument("--rgba", type=int, default=0, help="1: the primary leg fuses Frame.RGBA into the reconstruction kernel")
    ap.add_argument("--legs", default="dense,rgba_fused,dense_rgba_fused,mixed",
                    help="secondary video legs (comma separated; N=1 only): dense, rgba_fused, dense_rgba_fused, mixed; '' = none")
    ap.add_argument("--rgba-streams", type=int, default=-1, help="streams of the fused-RGBA legs (-1 = --streams; 0 = skip them)")
    ap.add_argument("--host-fed-seconds", type=float, default=2.0,
                    help="host-fed leg: pictures pushed through the staged submit from host threads for this many seconds "
                         "(N=1 only; PCIe inclusive, not `value`; 0 = skip, e.g. under a kernel trace: it launches the "
                         "reconstruction kernel on small batches)")
    ap.add_argument("--single-stream", type=int, default=1, help="1: the one-stream fused-RGBA leg of BASELINE config 3 (N=1 only)")
    ap.add_argument("--reference-benchmarks", type=int, default=1,
                    help="1: the reference's own micro-benchmarks through this path (copyMacroblock modes; DecodeVideo / DecodeAudio / "
                         "RGBA on the golden test files), N=1 only; also off with --single-stream 0 (the quick / traced runs)")
    ap.add_argument("--audio-tile", type=int, default=8,
                    help="second audio point: --audio-streams x this many streams (working set beyond the Infinity Cache; 1 = skip)")
    ap.add_argument("--audio-streams", type=int, default=256)
    ap.add_argument("--audio-frames", type=int, default=100)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--tile", type=int, default=0, help="mpeghip_video_set_tile_policy of every video leg: 0 = the library picks the "
                    "kernel instance per batch (the product's behaviour), 1 = the instance that transposes across lanes, 2 = the instance for dense units (for A/B runs)")
    ap.add_argument("--pin-numa", type=int, default=1, help="1: bind the rank to the cores of the NUMA node its GPU is attached to")
    ap.add_argument("--check", type=int, default=1, help="verify the final frames against the oracle (rank 0)")
    ap.add_argument("--share-devices", action="store_true",
                    help="allow ranks to share a physical GPU (functional runs of the N>1 path on a smaller box): without it a launch "
                         "with more ranks than distinct devices exits non-zero; with it the line reports n_gpus = the distinct devices")
    ap.add_argument("--gop-prewarm", type=int, default=1, help="1: whole untimed GOPs (40 ms of device work) in front of the --warmup steps (clock ramp)")
    ap.add_argument("--sif-streams", type=int, default=8192, help="streams of the SIF 352x240 leg (BASELINE config 2's geometry; N=1 only; 0 = skip)")
    ap.add_argument("--sidecar", default=str(ROOT / "bench_legs.json"),
                    help="where the FULL result goes (every leg with its prose: metric, sample, kernel, parity sentences); the one "
                         "JSON line on stdout is the compact form of it (< 7 500 characters); '' = do not write")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rendezvous only: start the ranks, form the gloo group, print {\"dry_launch\": true, \"ranks\": N} from rank 0 — "
                         "no GPU is touched (the CPU test of the N > 1 launch path)")
    return ap.parse_args(argv)


def cpu_baseline(args, seq):
    """Time the oracle (CPU restatement of the reference's noasm algorithm) on a bounded
    sample of the same workload: T host threads, one independent stream each."""
    from oracle import pyoracle
    from mpeg_amd.shard import effective_cores
    # one thread per core's worth of CPU time the process really gets (the cgroup quota, capped by the affinity mask), as the
    # library sizes its own pools: BENCH_r05 started 256 threads under a 16-core quota, which handicaps the baseline
    eff = effective_cores()
    threads = max(1, int(np.ceil(eff["effective_cores"])))
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
        # `cores` = the threads started = ceil(effective_cores): the quota if there is one, else the affinity mask
        "effective_cores": eff["effective_cores"], "cgroup_quota_cores": eff["cgroup_quota_cores"], "affinity_cpus": eff["affinity_cpus"],
        "sample": "oracle (C restatement of the reference's pure-Go noasm path, gcc -O2): %d host threads x 1 "
                  "1080p stream each over the bench GOP (%s profile), %d macroblocks in %.1f s; single thread: %.3g "
                  "macroblocks/s" % (threads, args.profile, nT, args.cpu_seconds * 0.75, r1),
        "single_thread": r1,
        # what the threads really got: a container with a CPU-time quota runs 256 threads on far fewer cores' worth of time
        "speedup_over_one_thread": rT / r1,
    }


def cpu_baseline_audio(args, seconds):
    """The other half of the metric on the host: the oracle's idct36 + synthWindow + scaling (oracle_desc.c: orc_synth_frames, the
    pure-Go arithmetic, no FMA) on config 4's samples — one thread, then one stream per host thread (ctypes releases the GIL)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from mpeg_amd import synth
    from oracle import pyoracle
    L = pyoracle.lib()
    from mpeg_amd.shard import effective_cores
    cores = max(1, int(np.ceil(effective_cores()["effective_cores"])))
    frames = 20
    smp = synth.audio_frames(min(cores, 64), frames)     # (streams beyond 64 reuse the samples: each has its own state and output)

    def run(stream, until):
        st, out, n = pyoracle.Synth(), np.empty((frames, 2304), np.float32), 0
        s = np.ascontiguousarray(smp[stream % len(smp)])
        while time.perf_counter() < until:
            L.orc_synth_frames(C.byref(st), s.ctypes.data_as(C.c_void_p), frames, 0, 0, out.ctypes.data_as(C.c_void_p))
            n += frames
        return n

    t0 = time.perf_counter()
    n1 = run(0, t0 + seconds * 0.3)
    r1 = n1 * 1152 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        nT = sum(ex.map(lambda i: run(i, t0 + seconds * 0.7), range(cores)))
    rT = nT * 1152 / (time.perf_counter() - t0)
    return {"value": rT, "unit": "MP2 stereo sample pairs/s", "cores": cores, "kind": "port", "single_thread": r1,
            "speedup_over_one_thread": rT / r1,
            "sample": "oracle (C restatement of audio.go:378-422, 492-772 + audio_noasm.go:8-38, gcc -O2, no FMA): %d host threads x 1 "
                      "stereo stream each, config 4's seeded sub-band samples, %d frames in %.1f s" % (cores, nT, seconds * 0.7)}


def cpu_baseline_test_mpg(seconds):
    """BASELINE config 1 as written (mpeg_test.go:463-491: BenchmarkDecodeVideo / BenchmarkDecodeAudio on testdata/test.mpg,
    frames and samples dropped) on the oracle, one host thread: demux + parse + reconstruct every video frame, then every audio
    frame, in a loop of at least 20 passes."""
    from oracle import pyoracle
    ps = ROOT / "tests" / "golden" / "test.mpg"
    if not ps.exists():
        return None
    data = ps.read_bytes()
    out = {"kind": "port", "cores": 1, "unit": "frames/s"}
    for what, packet, make in (("video", 0xE0, pyoracle.VideoDecoder), ("audio", 0xC0, pyoracle.AudioDecoder)):
        t0, loops, frames = time.perf_counter(), 0, 0
        while loops < 20 or time.perf_counter() - t0 < seconds / 2:
            es, _ = pyoracle.ps_extract(data, packet)
            dec = make(es)
            while dec.decode() is not None:
                frames += 1
            dec.close()
            loops += 1
        dt = time.perf_counter() - t0
        out[what + "_frames_per_s"] = frames / dt
        out[what + "_loops"] = loops
        out[what + "_frames_per_loop"] = frames // loops
    out["sample"] = ("oracle (C restatement of the reference's pure-Go path), ONE thread: tests/golden/test.mpg (160x120, MP2) demuxed "
                     "and decoded completely, %d video / %d audio passes" % (out["video_loops"], out["audio_loops"]))
    return out


def build_sequence(args, profile, rgba):
    """-> (seq, prime): the pictures of seq[:prime] run ONCE, untimed, before the warm-up; warm-up and timed steps cycle
    seq[prime:].  Typical: the whole decode-order GOP is cycled (prime = 0).  Dense (worst case per SURVEY §8(d)): the I
    picture only primes the frame stores; every warm-up and every timed step is a dense P picture (every block full, odd
    vectors in both axes, 1 635 algorithmic bytes per macroblock)."""
    from mpeg_amd import desc, synth
    seq = synth.generate_sequence(args.width, args.height, args.gop, profile=profile, rgba=rgba)
    if profile == "dense" or profile.startswith("mc_"):
        return seq[:1] + [s for s in seq[1:] if s.picture_type == desc.PIC_P], 1
    return seq, 0


def sources_sha256():
    """sha256 over the kernel sources the loaded libmpeghip.so was built from (mpeg_amd/csrc/*, sorted by name) and the
    compiler flags it was built with (mpeg_amd/_build.py): the PMC traffic figures under profiles/ carry the same hash of
    the build they were measured on."""
    import hashlib
    from mpeg_amd import _build
    h = hashlib.sha256()
    for f in sorted((ROOT / "mpeg_amd" / "csrc").iterdir()):
        if f.suffix in (".h", ".hip"):
            h.update(f.name.encode() + b"\0" + f.read_bytes())
    h.update(" ".join(_build.HIPCC_FLAGS).encode())
    return h.hexdigest()


def traffic_of(profile, rgba, streams, args):
    """HBM bytes per launch from the committed PMC runs (profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate passes, tools/gpu_traffic.sh): a figure of the named profile run, not measured in this process.
    -> (bytes, source, the run's kernel sources are the ones this process loaded)"""
    tp = ROOT / "profiles" / "pmc_traffic.json"
    if not tp.exists():
        return None, None, None
    # (keys: the 1080p workloads by their profile name; another geometry carries it: "typical_352x240")
    geom = "" if (args.width, args.height) == (1920, 1080) else "_%dx%d" % (args.width, args.height)
    try:
        j = json.loads(tp.read_text())
        if "current" in j:  # (a pointer to the profile set of the shipped sources, not a copy of it)
            j = json.loads((ROOT / "profiles" / j["current"]).read_text())
        t = j.get(profile + ("_rgba" if rgba else "") + geom, {})
        if t.get("streams") == streams:
            return t.get("hbm_bytes_per_launch"), t.get("source"), j.get("csrc_sha256") == sources_sha256()
    except Exception:
        pass
    return None, None, None


def video_leg(ctx, args, profile, rgba, streams, ranks=None, device_sync=None, steps=None, ramp_ms=0.0, geometry=None):
    """Upload the GOP for `streams` streams, warm up, time `--steps` steps.  With `ranks` the timed region is bracketed
    by barrier + device sync on both sides and the elapsed time is the MAX over ranks (the primary leg).
    ramp_ms: for legs whose warm-up steps take microseconds (one stream) — the same pictures run on a throw-away store for that
    long in front of the warm-up, so that the timed launches do not run on the clock ramp of a GPU that sat idle during the upload
    (the 1024-stream legs' 13 warm-up steps are 25-50 ms of work: they need none, profiles/round4_o_bench_steps104_warmup52.json)."""
    from mpeg_amd import abi
    steps = args.steps if steps is None else steps
    if geometry is not None:   # (a leg at another picture size: the SIF legs)
        args = argparse.Namespace(**{**vars(args), "width": geometry[0], "height": geometry[1]})
    seq, prime = build_sequence(args, profile, rgba)
    store = abi.VideoStore(ctx, args.width, args.height, streams)
    store.set_tile_policy(args.tile)
    batches = [store.upload(s.pics, s.mbs, s.coefs, replicate=streams) for s in seq]
    ctx.sync()
    device_bytes_per_picture = float(np.mean([b.device_bytes for b in batches]))  # one stream's picture in the device format
    order = []

    def step(i):
        k = prime + i if i < 0 else prime + i % (len(batches) - prime)  # (i = -prime .. -1: the priming pictures)
        b = batches[k]
        b.run()
        order.append(k)
        return b

    if ramp_ms > 0:
        scratch = abi.VideoStore(ctx, args.width, args.height, streams)
        scratch.set_tile_policy(args.tile)
        sb = [scratch.upload(s.pics, s.mbs, s.coefs, replicate=streams) for s in seq]
        done = 0.0
        while done < ramp_ms:
            ctx.timer_start()
            for _ in range(8):
                for b in sb:
                    b.run()
            done += max(ctx.timer_stop_ms(), 1e-3)
        # (the scratch store stays open until the leg is over: closing it would synchronise and idle the GPU again)
    cycle = len(batches) - prime
    for i in range(-prime, 0):
        step(i)
    prewarm_steps = 0
    if ramp_ms == 0 and args.gop_prewarm:  # (whatever --warmup is: 13 warm-up steps of a SIF leg are 7 ms)
        # The upload above left the GPU's compute clocks parked; W warm-up steps of 2 - 4 ms do not bring them back when W is small
        # (the driver's W = 5: 0.612 where W = 13 gives 0.617, profiles/round4_v_bench_repeatability.txt / round4_o_*).  Whole
        # GOPs, untimed, in front of the W warm-up steps until the device has worked for 40 ms (round 6: one GOP of SIF pictures
        # is 7 ms, one of the dense leg's 11 ms — those legs moved by 3 % from run to run; 1080p typical: two GOPs): the pictures
        # are part of `order`, so the oracle replays them too.
        warm_ms = 0.0
        while warm_ms < 40.0 and prewarm_steps < 8 * cycle:
            ctx.timer_start()
            for i in range(cycle):
                step(i)
            warm_ms += ctx.timer_stop_ms()
            prewarm_steps += cycle
    for i in range(args.warmup):
        step(i)
    acc = {"mbs": 0, "alg": 0, "ev_ms": 0.0}

    def timed_body():
        ctx.timer_start()
        for i in range(args.warmup, args.warmup + steps):
            b = step(i)
            acc["mbs"] += b.n_mbs
            acc["alg"] += b.alg_bytes
        acc["ev_ms"] = ctx.timer_stop_ms()  # HIP events on the stream the kernels run on

    if ranks is not None:
        elapsed = ranks.timed(timed_body, device_sync=device_sync)  # barrier + sync on both sides, MAX over ranks
        local_elapsed = ranks.last_local
    else:
        ctx.sync()
        t0 = time.perf_counter()
        timed_body()
        ctx.sync()
        elapsed = local_elapsed = time.perf_counter() - t0

    # parity at full size: every stream x 3 slots equal to the oracle's replay (device-side FNV-1a-64 per stream);
    # for the fused legs also the RGBA images of the first and the last stream
    check = None
    if args.check and (ranks is None or ranks.rank == 0):
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
        check = "bit-exact vs oracle on all %d streams x 3 slots after %d pictures" % (streams, len(order))
        if rgba and ok:
            probe = sorted({0, streams - 1})
            for slot in range(3):
                want = ref.read_rgba(0, slot)
                for st in probe:
                    ok &= bool(np.array_equal(np.asarray(store.read_rgba(st, slot)).reshape(-1), want.reshape(-1)))
            check += "; RGBA images bit-exact on streams %s x 3 slots" % probe
        ref.close()
        if not ok:
            raise SystemExit("bench: %s%s frames differ from the oracle — result invalid" % (profile, " (fused RGBA)" if rgba else ""))
        # every stream holds the same bytes so far: a per-stream base that is off by a stream would still hash right.  Give
        # far-apart streams their own reference content, run the predicted pictures of the same batches once more
        # (untimed), compare each of them with its own oracle replay
        if streams > 1:
            from mpeg_amd import desc
            from oracle import crosscheck
            ok, text = crosscheck.distinct_content_check(store, args.width, args.height, desc.geometry(args.width, args.height),
                                                         streams, seq, batches, rgba=bool(rgba))
            if not ok:
                raise SystemExit("bench: %s%s: %s — result invalid" % (profile, " (fused RGBA)" if rgba else "", text))
            check += "; " + text
    for b in batches:
        b.free()
    store.close()
    if ramp_ms > 0:
        for b in sb:
            b.free()
        scratch.close()
    launch_ms = acc["ev_ms"] / steps
    achieved = (acc["alg"] / steps) / (launch_ms * 1e-3) / 1e9
    traffic, source, matches = traffic_of(profile, rgba, streams, args)
    return {
        "seq": seq, "elapsed": elapsed, "local_elapsed": local_elapsed, "mbs": acc["mbs"], "gop_len": len(batches), "parity": check,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": source, "traffic_source_matches_build": matches, "kernel": KERNEL[bool(rgba)],
                     "alg_bytes_per_launch": acc["alg"] // steps, "avg_launch_ms": launch_ms},
        "steps": steps, "device_bytes_per_picture": device_bytes_per_picture,
        # untimed launches in front of the timed ones: the priming pictures, one whole GOP when --warmup is shorter than a GOP
        # (--gop-prewarm: the upload leaves the GPU's clocks parked), then the --warmup steps
        "untimed_prime_steps": prime, "untimed_prewarm_steps": prewarm_steps,
    }


def mixed_leg(ctx, args, streams, n_seeds=16):
    """What many concurrent streams look like: stream s decodes the GOP of seed s % 16, s % gop pictures ahead of stream 0 — I, P
    and B pictures of different streams, with different content, in ONE launch (mpeg_amd/mixed.py).  One resident batch per step
    of the GOP cycle (1024 different pictures each, validated and packed by the library's host packer at upload); parity: all
    streams x 3 slots, every distinct (seed, phase) combination against its own oracle replay."""
    from mpeg_amd import abi, mixed
    t0 = time.perf_counter()
    wl = mixed.MixedWorkload(args.width, args.height, streams, gop=args.gop, n_seeds=n_seeds, threads=min(16, os.cpu_count() or 1))
    store = abi.VideoStore(ctx, args.width, args.height, streams)
    store.set_tile_policy(args.tile)
    batches = []
    for t in range(args.gop):
        batches.append(store.upload(*wl.step_arrays(t)))
    ctx.sync()
    setup_s = time.perf_counter() - t0
    order = []

    def step(t):
        b = batches[t % args.gop]
        b.run()
        order.append(t)
        return b

    # Whole cycles, untimed, until the device has worked for 80 ms: every stream has decoded its references after the first, and
    # the setup above (seconds of host work) left the GPU's clocks parked — one cycle of 21 ms does not always bring them back
    # (the leg read 0.55 - 0.62 on two boxes of round 6 where its kernel runs at 0.69: gpurun_out/r6o, r6y; the primary leg has
    # --gop-prewarm for the same reason).  The pictures are part of `order`: the oracle replays them too.
    warm_ms, t_next = 0.0, 0
    while warm_ms < 80.0 and t_next < 8 * args.gop:
        ctx.timer_start()
        for t in range(t_next, t_next + args.gop):
            step(t)
        warm_ms += ctx.timer_stop_ms()
        t_next += args.gop
    ctx.sync()
    mbs = alg = 0
    w0 = time.perf_counter()
    ctx.timer_start()
    for t in range(t_next, t_next + args.steps):
        b = step(t)
        mbs += b.n_mbs
        alg += b.alg_bytes
    ev_ms = ctx.timer_stop_ms()
    elapsed = time.perf_counter() - w0
    parity = None
    if args.check:
        from oracle import mixedcheck
        ok, parity = mixedcheck.check(wl, store, order, threads=min(32, os.cpu_count() or 1))
        if not ok:
            raise SystemExit("bench: mixed leg: frames differ from the oracle — result invalid")
    for b in batches:
        b.free()
    store.close()
    launch_ms = ev_ms / args.steps
    achieved = (alg / args.steps) / (launch_ms * 1e-3) / 1e9
    return {"metric": "1080p macroblocks/sec, mixed: stream s at GOP phase s mod %d with the content of seed s mod %d — I, P and B "
                      "pictures of different streams in every launch" % (args.gop, n_seeds),
            "value": mbs / elapsed, "unit": "macroblocks/s", "streams": streams, "steps": args.steps,
            "ms_per_step": elapsed * 1e3 / args.steps, "realtime_1080p30_streams": mbs / elapsed / MB_PER_1080P30_STREAM,
            "distinct_seeds": n_seeds, "distinct_combinations": len(wl.combos()), "setup_s": setup_s, "untimed_warm_steps": t_next,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": KERNEL[False], "alg_bytes_per_launch": alg // args.steps, "avg_launch_ms": launch_ms},
            "parity": parity}


def secondary(leg, name, streams, args):
    return {"metric": "1080p macroblocks/sec, %s" % name, "value": leg["mbs"] / leg["elapsed"], "unit": "macroblocks/s",
            "streams": streams, "steps": leg["steps"], "ms_per_step": leg["elapsed"] * 1e3 / leg["steps"],
            "realtime_1080p30_streams": leg["mbs"] / leg["elapsed"] / MB_PER_1080P30_STREAM,
            "untimed_prewarm_steps": leg["untimed_prewarm_steps"], "roofline": leg["roofline"], "parity": leg["parity"]}


def audio_leg(ctx, args, streams, tile=1, fma=0, ranks=None, device_sync=None):
    """MP2 synthesis on `streams` x `tile` stereo streams, --audio-frames frames per launch.  tile > 1: the seeded samples
    of `streams` streams are uploaded `tile` times side by side (stream s = stream s mod `streams`): the working set of the
    timed launch then exceeds the 256 MB Infinity Cache, and every stream is still compared with the oracle.
    fma: 0 = multiply and add rounded separately in the window (the reference's pure-Go / SSE2 synthWindow, audio_noasm.go:8-38:
    golden hash 0xf1b7...), 1 = fused multiply-add (its amd64 AVX2 routine, audio_amd64.s: golden hash 0x50f3...)."""
    import ctypes as C
    from mpeg_amd import abi, desc, synth
    n, frames = streams * tile, args.audio_frames
    a = abi.AudioSynth(ctx, n, desc.AUDIO_FMA_WINDOW if fma else desc.AUDIO_FMA_NONE)
    smp = synth.audio_frames(streams, frames)
    d_s, d_o = a.device_buffers(frames, desc.AUDIO_F32N)
    for t in range(tile):
        a.upload(C.c_void_p(d_s.value + t * smp.nbytes), smp)
    a.synth_device(d_s, frames, desc.AUDIO_F32N, d_o)  # first launch from the zero state: checked below
    ctx.sync()
    aparity = None
    if args.check:
        from oracle import pyoracle
        want = pyoracle.OracleSynth(streams, 1 if fma else 0).synth(smp, desc.AUDIO_F32N).reshape(streams, -1)
        for t in range(tile):
            got = a.download(C.c_void_p(d_o.value + t * want.nbytes), want.size, desc.AUDIO_F32N).reshape(streams, -1)
            if not np.array_equal(want.view(np.uint32), got.view(np.uint32)):
                raise SystemExit("bench: audio samples differ from the oracle — result invalid")
        aparity = "bit-exact vs oracle (%s) on all %d streams x %d frames" % ("FMA window" if fma else "no-FMA", n, frames)
    # The oracle check above left the GPU idle for seconds and its clocks parked: a handful of sub-millisecond launches straight
    # after it are timed on the ramp (profiles/round4_n_audio_sustained_*.txt: the first 20 ms run 2-25 % slow, then the
    # duration is flat for seconds).  So: >= 40 ms of untimed launches, then >= 30 ms of timed ones.
    ctx.timer_start()
    for _ in range(2):
        a.synth_device(d_s, frames, desc.AUDIO_F32N, d_o)
    est = max(ctx.timer_stop_ms() / 2, 1e-3)
    for _ in range(int(np.ceil(40.0 / est))):
        a.synth_device(d_s, frames, desc.AUDIO_F32N, d_o)
    ctx.sync()
    world = ranks.world if ranks is not None else 1
    reps = int(min(max(np.ceil(30.0 / est), 5 if world == 1 else 40), 1000))
    if ranks is not None and world > 1:
        reps = int(max(ranks.gather(reps)))  # the same count on every rank
    ev = {}

    def timed_body():
        ctx.timer_start()
        for _ in range(reps):
            a.synth_device(d_s, frames, desc.AUDIO_F32N, d_o)
        ev["ms"] = ctx.timer_stop_ms()  # HIP events on the stream the kernel runs on

    if world > 1:  # every rank on its own streams between two barriers: MAX elapsed over ranks, as the video leg
        elapsed = ranks.timed(timed_body, device_sync=device_sync)
        per_rank = ranks.gather(n * frames * 1152 * reps / ranks.last_local)
    else:
        timed_body()
        elapsed, per_rank = ev["ms"] * 1e-3, None
    ams = ev["ms"] / reps
    abytes = n * frames * 18432
    traffic, source, matches = traffic_of("audio_%d" % n, False, n, args)
    out = {
        "metric": "MP2 stereo sample pairs/s",
        # N = 1: by the kernel's HIP events; N > 1: all ranks' sample pairs / the slowest rank's wall time between the barriers
        "value": n * frames * 1152 / (ams * 1e-3) if world == 1 else world * n * frames * 1152 * reps / elapsed,
        "n_gpus": getattr(args, "n_gpus", world), "ranks": world, "per_rank_value": per_rank, "launches_timed": reps,
        "streams": n, "frames_per_launch": frames, "ms_per_launch": ams,
        "realtime_streams_44k1": n * frames * 1152 / (ams * 1e-3) / 44100.0,
        "working_set_bytes": 2 * abytes,
        "roofline": {"bound": "hbm", "achieved": abytes / (ams * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": abytes / (ams * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source,
                     "traffic_source_matches_build": matches, "alg_bytes_per_launch": abytes,
                     "kernel": "audio_kernel<%s, F32N> (DCT-32 + polyphase window as a sliding register file, 4 waves per stream slice)" % ("true" if fma else "false")},
        "parity": aparity,
    }
    a.close()
    return out


def single_stream_leg(ctx, args):
    """BASELINE config 3 as written: ONE 1920x1080 stream on the GPU, IDCT + MC + YCbCr->RGBA fused, a picture per launch.
    8 160 macroblocks = 2 040 chunks = 2 040 four-wave workgroups of recon_wide_kernel (8 160 waves: the device's slots) on a 256-CU
    part: a latency figure, not a bandwidth one."""
    out = {"metric": "one %dx%d stream, one picture per launch, Frame.RGBA() fused (BASELINE config 3)" % (args.width, args.height)}
    for profile in ("typical", "dense"):
        leg = video_leg(ctx, args, profile, True, 1, steps=100, ramp_ms=40.0)
        r = leg["roofline"]
        out[profile] = {"us_per_picture": r["avg_launch_ms"] * 1e3, "pictures_per_s": 1e3 / r["avg_launch_ms"],
                        "macroblocks_per_s": leg["mbs"] / 100 / (r["avg_launch_ms"] * 1e-3), "achieved_GBps": r["achieved"],
                        "frac": r["frac"], "alg_bytes_per_launch": r["alg_bytes_per_launch"], "pictures_timed": 100,
                        "wall_us_per_picture": leg["elapsed"] * 1e4, "parity": leg["parity"]}
    return out


def sif_leg(ctx, args):
    """The other named resolution (BASELINE config 2: 352x240 SIF, 22 x 15 = 330 macroblocks per picture; SURVEY section 8(d)'s
    generator and seed): `--sif-streams` streams' typical GOP resident, one picture of every stream per launch — the roofline leg —
    and ONE SIF stream per launch (83 chunks: recon_wide_kernel's home ground, a latency figure).  mb_w = 22 is not a multiple of
    the chunk's 4 macroblocks: every row's last chunk carries on into the next row (a run for the plane stores all the same)."""
    geo, n = (352, 240), args.sif_streams
    leg = video_leg(ctx, args, "typical", False, n, geometry=geo)
    out = secondary(leg, "SIF 352x240 (BASELINE config 2's geometry), typical GOP, %d streams resident, one picture each per launch" % n, n, args)
    out["metric"] = "352x240 macroblocks/sec"
    del out["realtime_1080p30_streams"]
    out["realtime_sif30_streams"] = out["value"] / (330 * 30)
    out["pictures_per_s"] = out["value"] / 330
    # (500 launches: 100 of them are half a millisecond in all, and the figure then swings between 3.7 and 6.1 us from run to run on
    # one box — gpurun_out/r6d)
    one = video_leg(ctx, args, "typical", False, 1, steps=500, ramp_ms=40.0, geometry=geo)
    r = one["roofline"]
    out["single"] = {"us_per_picture": r["avg_launch_ms"] * 1e3, "pictures_per_s": 1e3 / r["avg_launch_ms"], "achieved_GBps": r["achieved"],
                     "frac": r["frac"], "alg_bytes_per_launch": r["alg_bytes_per_launch"], "pictures_timed": 500,
                     "wall_us_per_picture": one["elapsed"] * 1e6 / 500, "parity": one["parity"],
                     "what": "ONE 352x240 stream, one picture (330 macroblocks = 83 chunks) per launch"}
    return out


def reference_benchmarks(ctx, args, device):
    """The reference's own benchmarks through this path.  BenchmarkCopyMacroblock{Copy,Horiz,Vert,Bilin} (video_test.go:105-118:
    copyMacroblock with the vectors (0,0), (1,0), (0,1), (3,3)) as pictures of 1024 streams in which every macroblock is
    predicted with that vector and has no coded block; BenchmarkDecodeVideo / DecodeAudio / RGBA (mpeg_test.go:463-508) on
    testdata/test.mpg through the host stack (libmpeghost: demux + parse on one host thread, one submit per picture, every
    frame / sample block read back as the benchmark's loop receives them)."""
    out = {"copy_macroblock": {}}
    for mode in ("mc_copy", "mc_horiz", "mc_vert", "mc_bilin"):
        leg = video_leg(ctx, args, mode, False, args.streams)
        out["copy_macroblock"][mode[3:]] = {"macroblocks_per_s": leg["mbs"] / leg["elapsed"], "ns_per_macroblock": leg["elapsed"] / leg["mbs"] * 1e9,
                                            "frac": leg["roofline"]["frac"], "achieved_GBps": leg["roofline"]["achieved"],
                                            "avg_launch_ms": leg["roofline"]["avg_launch_ms"], "parity": leg["parity"]}
    import ctypes as C
    ps = ROOT / "tests" / "golden" / "test.mpg"
    so = ROOT / "mpeg_amd" / "libmpeghost.so"
    if ps.exists() and so.exists():
        H = C.CDLL(str(so))
        P = C.c_void_p
        H.mpeghost_device_create.restype, H.mpeghost_device_create.argtypes = P, [C.c_int]
        H.mpeghost_mpeg_open.restype, H.mpeghost_mpeg_open.argtypes = P, [P, C.c_char_p, C.c_size_t]
        H.mpeghost_mpeg_decode_video.restype, H.mpeghost_mpeg_decode_video.argtypes = C.c_int, [P, C.c_void_p]
        H.mpeghost_mpeg_decode_audio.restype, H.mpeghost_mpeg_decode_audio.argtypes = P, [P, C.POINTER(C.c_double)]
        H.mpeghost_mpeg_set_enabled.argtypes = [P, C.c_int, C.c_int]
        H.mpeghost_mpeg_set_loop.argtypes = [P, C.c_int]
        H.mpeghost_mpeg_close.argtypes = [P]
        H.mpeghost_device_destroy.argtypes = [P]
        H.mpeghost_video_open.restype, H.mpeghost_video_open.argtypes = P, [P, C.c_char_p, C.c_size_t]
        H.mpeghost_video_decode.restype, H.mpeghost_video_decode.argtypes = C.c_int, [P, C.c_void_p]
        H.mpeghost_video_rgba.restype, H.mpeghost_video_rgba.argtypes = P, [P]
        H.mpeghost_video_close.argtypes = [P]
        data = ps.read_bytes()
        dev = H.mpeghost_device_create(device)
        frame = (C.c_uint8 * 256)()   # mpeghost_frame (opaque here: time, sizes, plane pointers)
        def run(video, n):
            m = H.mpeghost_mpeg_open(dev, data, len(data))
            H.mpeghost_mpeg_set_loop(m, 1)
            H.mpeghost_mpeg_set_enabled(m, 1 if video else 0, 0 if video else 1)
            t = C.c_double()
            for _ in range(20):  # warm-up
                H.mpeghost_mpeg_decode_video(m, frame) if video else H.mpeghost_mpeg_decode_audio(m, C.byref(t))
            t0 = time.perf_counter()
            for _ in range(n):
                H.mpeghost_mpeg_decode_video(m, frame) if video else H.mpeghost_mpeg_decode_audio(m, C.byref(t))
            dt = time.perf_counter() - t0
            H.mpeghost_mpeg_close(m)
            return n / dt
        out["decode_video_test_mpg"] = {"pictures_per_s": run(True, 2000), "what": "MPEG.DecodeVideo in a loop, 160x120, planes read back"}
        out["decode_audio_test_mpg"] = {"frames_per_s": run(False, 2000), "what": "MPEG.DecodeAudio in a loop, 1152 stereo samples per frame, read back"}
        es = ROOT / "tests" / "golden" / "test.mpeg1video"
        if es.exists():
            ed = es.read_bytes()
            v = H.mpeghost_video_open(dev, ed, len(ed))
            if v and H.mpeghost_video_decode(v, frame) == 1:
                for _ in range(20):
                    H.mpeghost_video_rgba(v)
                t0 = time.perf_counter()
                for _ in range(2000):
                    H.mpeghost_video_rgba(v)
                out["rgba_test_mpeg1video"] = {"frames_per_s": 2000 / (time.perf_counter() - t0),
                                               "what": "Frame.RGBA() of one decoded 160x120 frame in a loop: conversion on the device + read-back"}
            if v:
                H.mpeghost_video_close(v)
        H.mpeghost_device_destroy(dev)
    return out


def host_fed_leg(args, prim, device, ranks=None):
    """The same typical pictures handed over by host threads through the staged submit — PCIe inclusive, NOT `value`.
    Headline: the DEVICE-PACKED stage (mpeghip_video_stage_begin_device: the host copies the ABI's arrays into pinned staging,
    pack_kernel validates and packs them in front of recon_kernel) with 8 putting threads; beside it the same with the
    pictures already in the staging buffers (a parser that writes in place: mpeghip_video_stage_map), and round 3's
    host-packed stages (validation + packing on the putting threads) for comparison."""
    import torch
    from mpeg_amd import desc
    from tools import hostbench
    cpus = os.cpu_count() or 1
    seq = prim["seq"]
    sec = args.host_fed_seconds
    w, h = args.width, args.height
    per_call = 128
    if ranks is not None and ranks.world > 1:
        # N > 1: every rank feeds its own GPU at the same time, 8 putting threads each, bound to its GPU's NUMA node (main)
        ranks.barrier()
        mine = hostbench.staged_submit_rate(device, w, h, seq, per_call, min(8, cpus), sec, sparse=True, device_pack=1)
        per_rank = ranks.gather(mine)
        mbpp = float(np.mean([len(s.mbs) for s in seq]))
        return {"metric": "1080p macroblocks/sec handed over by host threads through device-packed stages, all ranks at once "
                          "(8 putting threads per rank) — PCIe inclusive, NOT `value`",
                "value": sum(per_rank) * mbpp, "pictures_per_s": sum(per_rank), "per_rank_pictures_per_s": per_rank,
                "host_threads_per_rank": min(8, cpus), "pictures_per_call": per_call, "n_gpus": getattr(args, "n_gpus", ranks.world),
                "ranks": ranks.world}
    dev8 = hostbench.staged_submit_rate(device, w, h, seq, per_call, min(8, cpus), sec, sparse=True, device_pack=1)
    dev4 = hostbench.staged_submit_rate(device, w, h, seq, per_call, min(4, cpus), sec / 2, sparse=True, device_pack=1)
    dev16 = hostbench.staged_submit_rate(device, w, h, seq, per_call, min(16, cpus), sec / 2, sparse=True, device_pack=1)
    mid = seq[len(seq) // 2: len(seq) // 2 + 1]   # (a B picture of the GOP: the in-place mode cycles one picture)
    in_place = hostbench.staged_submit_rate(device, w, h, mid, per_call, min(8, cpus), sec / 2, sparse=True, device_pack=2)
    host32 = hostbench.staged_submit_rate(device, w, h, seq, 64, min(32, cpus), sec / 2, sparse=True)
    host16 = hostbench.staged_submit_rate(device, w, h, seq, 64, min(16, cpus), sec / 2, sparse=True)
    units32 = hostbench.staged_submit_rate(device, w, h, seq, 64, min(32, cpus), sec / 2, sparse=False)
    mb_per_pic = float(np.mean([len(s.mbs) for s in seq]))
    # bytes per picture on the wire: the ABI's arrays as they are (device-packed) / the device format (host-packed)
    abi_bytes = float(np.mean([32 * len(s.mbs) + 4 * len(desc.to_sparse(s.mbs, s.coefs)[1]) + 32 for s in seq]))
    mid_bytes = 32 * len(mid[0].mbs) + 4 * len(desc.to_sparse(mid[0].mbs, mid[0].coefs)[1]) + 32
    # what the link gives a bare copy from pinned memory (64 MB pieces, the size the staged submit sends)
    src = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
    dst = torch.empty(64 << 20, dtype=torch.uint8, device="cuda:%d" % device)
    h2d = 0.0
    for _ in range(4):  # (best of four: the first passes also fault the pinned pages in)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(16):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        h2d = max(h2d, 16 * (64 << 20) / (time.perf_counter() - t0) / 1e9)
    del src, dst
    return {"metric": "1080p macroblocks/sec handed over by host threads through a DEVICE-PACKED stage (mpeghip_video_stage_begin_device: "
                      "the host copies the ABI's arrays — descriptors + the parser's sparse words — into pinned staging, pack_kernel "
                      "validates and packs them on the GPU, recon_kernel reconstructs) — PCIe inclusive, NOT `value`",
            "value": dev8 * mb_per_pic, "pictures_per_s": dev8, "host_threads": min(8, cpus), "pictures_per_call": per_call,
            "pictures_per_s_4_threads": dev4, "pictures_per_s_16_threads": dev16,
            "realtime_1080p30_streams": dev8 * mb_per_pic / MB_PER_1080P30_STREAM,
            "abi_bytes_per_picture": abi_bytes, "pcie_GBps_used": dev8 * abi_bytes / 1e9, "pcie_h2d_GBps_bare_copy": h2d,
            "pcie_frac_of_bare_copy": dev8 * abi_bytes / 1e9 / h2d if h2d else None,
            "in_place": {"pictures_per_s": in_place, "pcie_GBps_used": in_place * mid_bytes / 1e9, "bytes_per_picture": mid_bytes,
                         "what": "the pictures already in the pinned staging buffers (mpeghip_video_stage_map: a parser writes them "
                                 "there): commit + PCIe + pack_kernel + recon_kernel, no host work per picture"},
            "host_packed": {"pictures_per_s_32_threads": host32, "pictures_per_s_16_threads": host16, "pictures_per_s_unit_form_32_threads": units32,
                            "pictures_per_call": 64, "device_format_bytes_per_picture": prim["device_bytes_per_picture"],
                            "pcie_GBps_used": host32 * prim["device_bytes_per_picture"] / 1e9,
                            "what": "round 3's hand-over: validation + packing into the device format on the putting threads"},
            "note": "the on-device `value` is %.0f x this: no PCIe link can carry what the kernel reconstructs (BASELINE.md)" %
                    (prim["mbs"] / prim["elapsed"] / (dev8 * mb_per_pic))}


def host_parsed_leg(args, device, streams=256, threads=16, gop=7, groups=6):
    """Parse-inclusive: a written 1080p elementary stream (tests/mpeg1_writer.py: the `natural` level mix, coefficients as
    Table B.5 run / level codes, escapes where the table has none) through the product's whole host stack — mpeg::VideoBatch:
    bitstream parse on a pool of host threads -> the parser's sparse pictures -> device-packed staged commits -> pack_kernel +
    recon_kernel; frames stay on the device.  What ONE GPU's host side delivers from bitstreams, NOT `value`."""
    import ctypes as C
    sys.path.insert(0, str(ROOT / "tests"))
    import mpeg1_writer
    from mpeg_amd import synth
    so = ROOT / "mpeg_amd" / "libmpeghost.so"
    if not so.exists():
        return None
    seq = synth.generate_sequence(args.width, args.height, gop, seed=0x5a, profile="natural")
    es = mpeg1_writer.write_sequence(args.width, args.height, seq, repeat=groups)
    H = C.CDLL(str(so))
    P = C.c_void_p
    H.mpeghost_device_create.restype, H.mpeghost_device_create.argtypes = P, [C.c_int]
    H.mpeghost_device_destroy.argtypes = [P]
    H.mpeghost_batch_open.restype, H.mpeghost_batch_open.argtypes = P, [P, C.c_uint32]
    H.mpeghost_batch_close.argtypes = [P]
    H.mpeghost_batch_add_stream.restype, H.mpeghost_batch_add_stream.argtypes = C.c_int, [P, C.c_char_p, C.c_size_t]
    H.mpeghost_batch_decode_all.restype, H.mpeghost_batch_decode_all.argtypes = C.c_int, [P, C.c_int]
    H.mpeghost_batch_set_threads.argtypes = [P, C.c_uint32]
    H.mpeghost_batch_set_device_pack.argtypes = [P, C.c_int]
    H.mpeghost_batch_sync.restype, H.mpeghost_batch_sync.argtypes = C.c_int, [P]
    H.mpeghost_batch_phase_seconds.argtypes = [P, C.POINTER(C.c_double * 4)]
    H.mpeghost_batch_counters.argtypes = [P, C.POINTER(C.c_uint64 * 2)]
    H.mpeghost_last_error.restype = C.c_char_p
    H.mpeghost_batch_threads.restype, H.mpeghost_batch_threads.argtypes = C.c_uint32, [P]
    H.mpeghost_effective_cores.restype = C.c_double
    # the pool is sized by the CPU time the process gets (the cgroup quota), not by the request: BENCH_r04 ran 64 threads under a
    # ~10-core quota 26 % SLOWER than 16.  `threads` below is the request; every run reports what the pool became.
    eff = float(H.mpeghost_effective_cores())
    threads = max(1, min(threads, os.cpu_count() or 1))
    dev = H.mpeghost_device_create(device)
    out = {"metric": "1080p pictures/s from BITSTREAMS: %d streams of a written 1080p stream (natural level mix, Table B.5 codes; %d "
                     "pictures, %.0f kB per picture) parsed on %d host threads, handed over as device-packed staged commits, "
                     "reconstructed — NOT `value`" % (streams, gop * groups, len(es) / (gop * groups) / 1e3, threads),
           "streams": streams, "parse_threads_requested": threads, "effective_cores": eff,
           "pictures_per_stream": gop * groups, "stream_bytes_per_picture": len(es) / (gop * groups)}
    # (name, device-side packing, parse threads, streams).  A tick parses ONE picture of every stream: with few streams per thread
    # a round is short and its hand-over (stage begin, commit, the wait for the slowest thread) weighs — BENCH_r04 / round 5:
    # 64 streams on 16 threads 7 800 pictures/s, 256 streams 12 800.  The headline arm has 16 streams per thread; the few-streams
    # arm shows the other end.
    H.mpeghost_batch_device_pack.restype, H.mpeghost_batch_device_pack.argtypes = C.c_int, [P]
    # the first arm is the product's DEFAULT hand-over (nothing set: VideoBatch's own choice — device-packed since round 6, with
    # the verdict of every round asked for before the next one parses); the key says which form that was
    for name, device_pack, nthreads, nstreams in (("default", None, threads, streams), ("host_packed", 0, threads, streams),
                                                  ("device_packed_few_streams", 1, threads, max(threads, streams // 4))):
        b = H.mpeghost_batch_open(dev, nstreams)
        if not b:
            raise SystemExit("bench: host_parsed: %s" % H.mpeghost_last_error().decode())
        H.mpeghost_batch_set_threads(b, nthreads)
        nthreads = int(H.mpeghost_batch_threads(b))     # (what the request became: never more than the quota, rounded up)
        if device_pack is not None:
            H.mpeghost_batch_set_device_pack(b, device_pack)
        elif name == "default":
            out["default_arm"] = name = "device_packed" if H.mpeghost_batch_device_pack(b) else "host_packed_default"
        for _ in range(nstreams):
            if H.mpeghost_batch_add_stream(b, es, len(es)) < 0:
                raise SystemExit("bench: host_parsed: %s" % H.mpeghost_last_error().decode())
        t0 = time.perf_counter()
        while H.mpeghost_batch_decode_all(b, 0) > 0:
            pass
        if H.mpeghost_batch_sync(b) != 0:
            raise SystemExit("bench: host_parsed: %s" % H.mpeghost_last_error().decode())
        dt = time.perf_counter() - t0
        ph, cn = (C.c_double * 4)(), (C.c_uint64 * 2)()
        H.mpeghost_batch_phase_seconds(b, C.byref(ph))
        H.mpeghost_batch_counters(b, C.byref(cn))
        H.mpeghost_batch_close(b)
        pictures = int(cn[1])
        out[name] = {"pictures_per_s": pictures / dt, "pictures": pictures, "seconds": dt, "device_calls": int(cn[0]), "parse_threads": nthreads,
                     "streams": nstreams,
                     # wall time of the parse rounds x the cores that worked on them: min(threads, the quota) — under a quota
                     # of 10 cores 16 threads get 10 cores' worth of time, and "x threads" overstated the cost by 1.6 (BENCH_r04's
                     # 1.49 ms against the 1.03 ms a free core needs)
                     "ms_parse_per_picture_per_core": ph[0] * 1e3 * min(nthreads, eff) / max(pictures, 1),
                     "ms_parse_per_picture_per_thread": ph[0] * 1e3 * nthreads / max(pictures, 1),
                     "wall_seconds": {"parse_rounds": ph[0], "stage_begin": ph[1], "puts": ph[2], "commits": ph[3]}}
    H.mpeghost_device_destroy(dev)
    out["value"] = out[out["default_arm"]]["pictures_per_s"]
    out["realtime_1080p30_streams"] = out["value"] / 30.0
    return out


def audio_host_parsed_leg(args, device, streams=256, threads=16):
    """MP2 from BITSTREAMS: `streams` copies of tests/golden/test.mp2 (the reference's own audio fixture: 355 frames) through
    mpeg::AudioBatch — every tick parses one frame of every stream on the pool (header, allocation, scale factors,
    requantisation: audio.go:163-376, 429-490), synthesises all of them with ONE device call and hands the samples back to the
    host.  Frames/s with 1 parse thread and with `threads`: what one GPU's host side delivers from MP2 bitstreams, NOT `value`."""
    import ctypes as C
    so = ROOT / "mpeg_amd" / "libmpeghost.so"
    mp2 = ROOT / "tests" / "golden" / "test.mp2"
    if not so.exists() or not mp2.exists():
        return None
    H = C.CDLL(str(so))
    P = C.c_void_p
    H.mpeghost_device_create.restype, H.mpeghost_device_create.argtypes = P, [C.c_int]
    H.mpeghost_device_destroy.argtypes = [P]
    H.mpeghost_audio_batch_open.restype, H.mpeghost_audio_batch_open.argtypes = P, [P, C.c_uint32, C.c_int, C.c_int]
    H.mpeghost_audio_batch_close.argtypes = [P]
    H.mpeghost_audio_batch_add_stream.restype, H.mpeghost_audio_batch_add_stream.argtypes = C.c_int, [P, C.c_char_p, C.c_size_t]
    H.mpeghost_audio_batch_decode_all.restype, H.mpeghost_audio_batch_decode_all.argtypes = C.c_int, [P]
    H.mpeghost_audio_batch_set_threads.argtypes = [P, C.c_uint32]
    H.mpeghost_audio_batch_device_calls.restype, H.mpeghost_audio_batch_device_calls.argtypes = C.c_uint64, [P]
    H.mpeghost_last_error.restype = C.c_char_p
    data = mp2.read_bytes()
    threads = max(1, min(threads, os.cpu_count() or 1))
    dev = H.mpeghost_device_create(device)
    out = {"metric": "MP2 frames/s from BITSTREAMS: %d copies of tests/golden/test.mp2 parsed by mpeg::AudioBatch, one synthesis call per "
                     "tick (one frame of every stream), samples back on the host — NOT `value`" % streams, "streams": streams}
    for name, n in (("one_thread", 1), ("%d_threads" % threads, threads)):
        b = H.mpeghost_audio_batch_open(dev, streams, 0, 0)
        if not b:
            raise SystemExit("bench: audio_host_parsed: %s" % H.mpeghost_last_error().decode())
        H.mpeghost_audio_batch_set_threads(b, n)
        for _ in range(streams):
            if H.mpeghost_audio_batch_add_stream(b, data, len(data)) < 0:
                raise SystemExit("bench: audio_host_parsed: %s" % H.mpeghost_last_error().decode())
        frames, t0 = 0, time.perf_counter()
        while True:
            k = H.mpeghost_audio_batch_decode_all(b)
            if k < 0:
                raise SystemExit("bench: audio_host_parsed: %s" % H.mpeghost_last_error().decode())
            if k == 0:
                break
            frames += k
        dt = time.perf_counter() - t0
        calls = int(H.mpeghost_audio_batch_device_calls(b))
        H.mpeghost_audio_batch_close(b)
        out[name] = {"parse_threads": n, "frames": frames, "seconds": dt, "frames_per_s": frames / dt, "sample_pairs_per_s": frames * 1152 / dt,
                     "device_calls": calls, "ms_per_tick": dt * 1e3 / max(calls, 1)}
    H.mpeghost_device_destroy(dev)
    best = out["%d_threads" % threads]
    out["value"] = best["frames_per_s"]
    out["realtime_streams_44k1"] = best["sample_pairs_per_s"] / 44100.0
    return out



COMPACT_LIMIT = 7500   # characters: the driver's record keeps a tail of ~8 000


def _r(x, digits=5):
    """Numbers of the compact line: `digits` significant digits (floats), everything else as it is."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    return x


def _leg(full, tag, value_key="value"):
    """One leg of the compact line: value, frac of the HBM roofline, ms per launch, algorithmic and counter bytes per launch,
    parity as a boolean (a leg whose frames differ from the oracle never gets here: bench.py exits), a short tag."""
    if not full:
        return None
    r = full.get("roofline") or {}
    out = {"tag": tag, "value": full.get(value_key), "frac": r.get("frac", full.get("frac")),
           "ms": r.get("avg_launch_ms", full.get("ms_per_launch")), "alg_bytes": r.get("alg_bytes_per_launch"), "traffic": r.get("traffic"),
           "parity_ok": bool(full.get("parity")) if "parity" in full else None}
    if r.get("traffic") is not None:
        out["traffic_matches_build"] = r.get("traffic_source_matches_build")
    if "streams" in full:
        out["streams"] = full["streams"]
    return {k: v for k, v in out.items() if v is not None}


def compact_line(full):
    """The ONE line on stdout: the contract's fields + roofline + cpu_baseline + every leg as numbers and booleans — the prose
    (metric sentences, samples, kernel descriptions, parity sentences, traffic sources) lives in the sidecar (--sidecar,
    bench_legs.json) and under profiles/.  Order of the extra keys: the second half of BASELINE's metric (audio, config 4, no FMA)
    and the worst-case / fused video legs first.  Stays under COMPACT_LIMIT characters (tests/test_bench_line.py)."""
    roof = full.get("roofline") or {}
    cpu = full.get("cpu_baseline")
    cfg = full.get("config") or {}
    line = {
        "metric": full["metric"], "value": full["value"], "unit": full["unit"], "n_gpus": full["n_gpus"], "ranks": full.get("ranks"),
        "steps": full["steps"], "warmup": full["warmup"], "ms_per_step": full["ms_per_step"], "higher_is_better": True,
        "scaling": full["scaling"], "vs_baseline": full.get("vs_baseline"), "dtype": full["dtype"], "data": full["data"],
        "config": {"workload": cfg.get("workload"), "streams_per_gpu": cfg.get("streams_per_gpu"),
                   "macroblocks_per_step_per_gpu": cfg.get("macroblocks_per_step_per_gpu"), "profile": cfg.get("profile"),
                   "rgba_fused": cfg.get("rgba_fused"), "sharding": "by stream, no collective", "devices": cfg.get("devices"),
                   "devices_shared": cfg.get("devices_shared"), "untimed_prewarm_steps": cfg.get("untimed_prewarm_steps")},
        "realtime_1080p30_streams": full.get("realtime_1080p30_streams"), "per_rank_value": full.get("per_rank_value"),
        "roofline": {"bound": roof.get("bound"), "achieved": roof.get("achieved"), "peak": roof.get("peak"), "unit": roof.get("unit"),
                     "frac": roof.get("frac"), "traffic": roof.get("traffic"), "traffic_matches_build": roof.get("traffic_source_matches_build"),
                     "alg_bytes_per_launch": roof.get("alg_bytes_per_launch"), "avg_launch_ms": roof.get("avg_launch_ms"),
                     "kernel": "recon_kernel<1,%s,T>" % ("true" if cfg.get("rgba_fused") else "false")},
        "parity_ok": bool(full.get("parity")),
    }
    if cpu:
        c = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
             "sample": "oracle (C port of the reference's noasm path, gcc -O2), %d threads x 1 stream, bench GOP, ~%.0f s" %
                       (cpu["cores"], 0.75 * full.get("cpu_seconds", 12.0)),
             "effective_cores": cpu.get("effective_cores"), "single_thread": cpu.get("single_thread")}
        a = cpu.get("audio")
        if a:
            c["audio"] = {"value": a["value"], "unit": "sample pairs/s", "cores": a["cores"], "single_thread": a.get("single_thread")}
        t = cpu.get("test_mpg")
        if t:
            c["test_mpg"] = {"video_frames_per_s": t.get("video_frames_per_s"), "audio_frames_per_s": t.get("audio_frames_per_s"), "cores": 1}
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    # ---- the legs
    au = full.get("audio") or {}
    line["audio"] = _leg(full.get("audio"), "%sMP2 synthesis, %s streams x %s frames per launch, no FMA; sample pairs/s" %
                         ("config 4: " if au.get("streams") == 256 else "", au.get("streams"), au.get("frames_per_launch")))
    if line["audio"] and full["audio"].get("ranks", 1) > 1:
        line["audio"]["per_rank_value"] = full["audio"].get("per_rank_value")
    line["dense"] = _leg(full.get("dense"), "1080p dense worst case, MB/s")
    line["dense_rgba_fused"] = _leg(full.get("dense_rgba_fused"), "1080p dense + Frame.RGBA fused, MB/s")
    line["rgba_fused"] = _leg(full.get("rgba_fused"), "1080p typical + Frame.RGBA fused, MB/s")
    line["mixed"] = _leg(full.get("mixed"), "1080p, stream s at its own GOP phase/content, MB/s")
    sif = full.get("sif")
    if sif:
        line["sif"] = _leg(sif, "352x240 typical GOP, MB/s")
        line["sif"]["pictures_per_s"] = sif.get("pictures_per_s")
        one = sif.get("single") or {}
        line["sif_single"] = {"tag": "ONE 352x240 stream, one picture per launch", "us_per_picture": one.get("us_per_picture"),
                              "frac": one.get("frac"), "parity_ok": bool(one.get("parity"))}
    line["audio_large"] = _leg(full.get("audio_large"), "2048 streams (beyond the Infinity Cache), sample pairs/s")
    line["audio_fma_window"] = _leg(full.get("audio_fma_window"), "config4 with the reference's AVX2 FMA window arithmetic")
    single = full.get("single_stream")
    if single:
        line["single_stream"] = {"tag": "ONE 1080p stream, picture per launch, RGBA fused (config 3)"}
        for k in ("typical", "dense"):
            v = single.get(k) or {}
            line["single_stream"][k] = {"us_per_picture": v.get("us_per_picture"), "frac": v.get("frac"), "parity_ok": bool(v.get("parity"))}
    rb = full.get("reference_benchmarks")
    if rb:
        c = {"copy_macroblock_frac": {k: v.get("frac") for k, v in (rb.get("copy_macroblock") or {}).items()},
             "copy_macroblock_parity_ok": all(bool(v.get("parity")) for v in (rb.get("copy_macroblock") or {}).values())}
        for k, f in (("decode_video_test_mpg", "pictures_per_s"), ("decode_audio_test_mpg", "frames_per_s"), ("rgba_test_mpeg1video", "frames_per_s")):
            if rb.get(k):
                c[k] = rb[k].get(f)
        line["reference_benchmarks"] = c
    hf = full.get("host_fed")
    if hf:
        line["host_fed"] = {"tag": "device-packed stages from host threads, PCIe inclusive, NOT value", "value": hf.get("value"),
                            "pictures_per_s": hf.get("pictures_per_s"), "pcie_frac_of_bare_copy": hf.get("pcie_frac_of_bare_copy"),
                            "in_place_pictures_per_s": (hf.get("in_place") or {}).get("pictures_per_s"),
                            "host_packed_pictures_per_s": (hf.get("host_packed") or {}).get("pictures_per_s_32_threads"),
                            "per_rank_pictures_per_s": hf.get("per_rank_pictures_per_s")}
        line["host_fed"] = {k: v for k, v in line["host_fed"].items() if v is not None}
    hp = full.get("host_parsed")
    if hp:
        line["host_parsed"] = {"tag": "1080p pictures/s from BITSTREAMS (parse threads -> stages -> GPU), NOT value", "value": hp.get("value"),
                               "default_arm": hp.get("default_arm"), "effective_cores": hp.get("effective_cores")}
        for k in ("device_packed", "host_packed", "device_packed_few_streams"):
            if hp.get(k):
                line["host_parsed"][k] = {"pictures_per_s": hp[k]["pictures_per_s"], "parse_threads": hp[k]["parse_threads"],
                                          "ms_parse_per_picture_per_core": hp[k]["ms_parse_per_picture_per_core"]}
    ap = full.get("audio_host_parsed")
    if ap:
        line["audio_host_parsed"] = {"tag": "MP2 frames/s from BITSTREAMS, samples back on the host, NOT value", "value": ap.get("value"),
                                     "one_thread": (ap.get("one_thread") or {}).get("frames_per_s")}
    line["csrc_sha256"] = (full.get("csrc_sha256") or "")[:16]
    line["sidecar"] = full.get("sidecar")
    line = _r({k: v for k, v in line.items() if v is not None or k in ("vs_baseline", "cpu_baseline")})
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= COMPACT_LIMIT:   # (cannot happen with the legs above; if a future leg grows: drop the tags, never the numbers)
        for v in line.values():
            if isinstance(v, dict):
                v.pop("tag", None)
    return line


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves — the command the driver's contract names
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same argv>)
    on a free port.  Rank 0 prints the ONE JSON line, the other ranks print nothing; the exit code is the launcher's (non-zero
    if any rank failed — e.g. the device census refusing ranks that share a GPU)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")   # (what the launcher would set itself, with a notice on stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    return subprocess.call(cmd, env=env)


def dry_launch(args):
    """--dry-launch: the ranks rendezvous over gloo and report who is there; nothing else runs (no GPU needed)."""
    from mpeg_amd.shard import Ranks
    ranks = Ranks(backend="gloo")
    who = ranks.gather_object({"rank": ranks.rank, "local_rank": ranks.local_rank, "pid": os.getpid()})
    total = ranks.sum(1.0)
    ranks.barrier()
    if ranks.rank == 0:
        print(json.dumps({"dry_launch": True, "gpus_asked": args.gpus, "ranks": ranks.world, "ranks_counted": int(total),
                          "local_ranks": [w["local_rank"] for w in who], "distinct_pids": len({w["pid"] for w in who})}))
    ranks.close()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, argv))
    if args.dry_launch:
        return dry_launch(args)
    import torch

    from mpeg_amd.shard import Ranks
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the product has no CPU path")
    n_dev = torch.cuda.device_count()
    local_rank %= n_dev  # (more ranks than visible devices: they share — refused below unless --share-devices)
    torch.cuda.set_device(local_rank)
    ranks = Ranks(backend="gloo")  # control plane only: barrier + reductions of timings (no collective on the data path)
    world, rank = ranks.world, ranks.rank
    if args.gpus != world and rank == 0:  # (a launcher's WORLD_SIZE is what runs; the line reports ranks and distinct devices)
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: running %d" % (args.gpus, world, world), file=sys.stderr)

    from mpeg_amd import abi
    from mpeg_amd.shard import device_census

    tstream = torch.cuda.Stream(device=local_rank)
    ctx = abi.Context(local_rank, tstream.cuda_stream)
    # which PHYSICAL devices the ranks are on (PCI addresses: ordinals are per process): n_gpus of every line below is the number
    # of distinct ones, and a launch whose ranks share a device is refused unless --share-devices says it is meant
    try:
        census = device_census(ranks.gather_object(ctx.pci_bus_id()), args.share_devices)
    except ValueError as e:
        ctx.close()
        ranks.close()
        raise SystemExit("bench.py: %s" % e)
    args.n_gpus = census["n_gpus"]
    # one process per GPU: this rank's host threads (staged puts of the host-fed leg, the CPU baseline) run on the socket
    # its GPU is attached to
    from mpeg_amd.shard import pin_to_node
    numa = {"node": ctx.numa_node(), "cpus_bound": 0}
    all_cpus = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    if args.pin_numa:
        numa["cpus_bound"] = pin_to_node(numa["node"])

    # ---- primary leg (per rank: its own streams, same seeded GOP)
    prim = video_leg(ctx, args, args.profile, bool(args.rgba), args.streams, ranks=ranks, device_sync=torch.cuda.synchronize)
    per_rank = ranks.gather(prim["mbs"] / prim["local_elapsed"])  # each rank's own rate: contention between ranks shows here

    legs = {}
    alone = rank == 0 and world == 1
    if alone:
        rs = args.streams if args.rgba_streams < 0 else args.rgba_streams
        for name in [x for x in args.legs.split(",") if x]:
            if name == "dense" and not (args.profile == "dense" and not args.rgba):
                legs["dense"] = secondary(video_leg(ctx, args, "dense", False, args.streams),
                                          "dense worst case (every block full, odd vectors; every timed step a P picture, the I picture only primes the stores)", args.streams, args)
            elif name == "rgba_fused" and rs > 0 and not (args.profile == "typical" and args.rgba):
                legs["rgba_fused"] = secondary(video_leg(ctx, args, "typical", True, rs),
                                               "Frame.RGBA() of every picture fused into the reconstruction kernel", rs, args)
            elif name == "mixed":
                legs["mixed"] = mixed_leg(ctx, args, args.streams)
            elif name == "dense_rgba_fused" and rs > 0 and not (args.profile == "dense" and args.rgba):
                legs["dense_rgba_fused"] = secondary(video_leg(ctx, args, "dense", True, rs),
                                                     "dense worst case with Frame.RGBA() fused", rs, args)

    # the other half of the metric: every rank synthesises its own --audio-streams streams (MAX time over ranks at N > 1)
    audio = audio_leg(ctx, args, args.audio_streams, ranks=ranks, device_sync=torch.cuda.synchronize) if args.audio_streams > 0 else None
    audio_large = None
    if args.audio_streams > 0 and alone and args.audio_tile > 1:  # beyond the Infinity Cache: config 4's working set is about its size
        audio_large = audio_leg(ctx, args, args.audio_streams, args.audio_tile)
    # the reference's other arithmetic: its amd64 AVX2 window routine uses fused multiply-adds (what it runs on any recent x86)
    audio_fma = audio_leg(ctx, args, args.audio_streams, fma=1) if args.audio_streams > 0 and alone else None
    single = single_stream_leg(ctx, args) if alone and args.single_stream else None
    sif = sif_leg(ctx, args) if alone and args.single_stream and args.sif_streams > 0 else None
    ref_bench = reference_benchmarks(ctx, args, local_rank) if alone and args.reference_benchmarks and args.single_stream else None

    # ---- host-fed rate (NOT `value`): the same pictures handed over by host threads through the staged submit,
    # i.e. validation + packing into the device format on the host, PCIe, reconstruction on the device
    host_fed = None
    if args.host_fed_seconds > 0:
        host_fed = host_fed_leg(args, prim, local_rank, ranks)

    host_parsed = host_parsed_leg(args, local_rank) if args.host_fed_seconds > 0 and alone and args.single_stream else None
    audio_host_parsed = audio_host_parsed_leg(args, local_rank) if args.host_fed_seconds > 0 and alone and args.single_stream and args.audio_streams > 0 else None

    if all_cpus is not None and numa["cpus_bound"]:
        os.sched_setaffinity(0, all_cpus)  # the CPU baseline is the whole host's: every core of both sockets
    # the CPU baselines: rank 0 only, also at N > 1 (after the timed legs; the other ranks wait at the closing barrier)
    cpu = None
    if args.cpu_seconds > 0 and rank == 0:
        cpu = cpu_baseline(args, prim["seq"])
        cpu["audio"] = cpu_baseline_audio(args, max(2.0, args.cpu_seconds / 3))
        cpu["test_mpg"] = cpu_baseline_test_mpg(max(2.0, args.cpu_seconds / 3))
    ranks.barrier()

    if rank == 0:
        total_mbs = prim["mbs"] * world
        value = total_mbs / prim["elapsed"]
        line = {
            "metric": "1080p macroblocks/sec", "value": value, "unit": "macroblocks/s",
            "n_gpus": census["n_gpus"], "ranks": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": prim["elapsed"] * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 pixels / int16 levels / int32 IDCT", "data": "synthetic",
            "config": {"workload": "%d independent %dx%d MPEG-1 streams per GPU, one picture each per step, decode-order "
                                   "GOP of %d pictures (%s macroblock mix%s), descriptors resident in HBM" %
                                   (args.streams, args.width, args.height, prim["gop_len"], args.profile,
                                    ", fused RGBA" if args.rgba else ""),
                       "streams_per_gpu": args.streams, "macroblocks_per_step_per_gpu": prim["mbs"] // args.steps,
                       "profile": args.profile, "rgba_fused": bool(args.rgba), "sharding": "by stream, no collective; control plane gloo",
                       "host_numa": numa,
                       # the physical device of every rank (PCI address, rank order); n_gpus = the distinct ones
                       "devices": census["devices"], "devices_shared": census["shared"],
                       # launches in front of the timed ones that `warmup` does not count (clock ramp: DESIGN.md section 5.0)
                       "untimed_prewarm_steps": prim["untimed_prewarm_steps"], "untimed_prime_steps": prim["untimed_prime_steps"]},
            "realtime_1080p30_streams": value / MB_PER_1080P30_STREAM,
            "per_rank_value": per_rank,
            "roofline": prim["roofline"],
            "cpu_baseline": cpu,
            "audio": audio,
            "dense": legs.get("dense"),
            "dense_rgba_fused": legs.get("dense_rgba_fused"),
            "rgba_fused": legs.get("rgba_fused"),
            "mixed": legs.get("mixed"),
            "sif": sif,
            "audio_large": audio_large,
            "audio_fma_window": audio_fma,
            "single_stream": single,
            "reference_benchmarks": ref_bench,
            "host_fed": host_fed,
            "host_parsed": host_parsed,
            "audio_host_parsed": audio_host_parsed,
            "parity": prim["parity"],
            "cpu_seconds": args.cpu_seconds,
        }
        line["csrc_sha256"] = sources_sha256()
        if args.sidecar:
            line["sidecar"] = os.path.relpath(args.sidecar, ROOT) if str(args.sidecar).startswith(str(ROOT)) else str(args.sidecar)
            try:   # the FULL result (every leg with its prose) next to the script; stdout carries the compact form
                Path(args.sidecar).write_text(json.dumps(line, indent=1) + "\n")
            except OSError as e:
                line["sidecar"] = None
                print("bench.py: sidecar not written: %s" % e, file=sys.stderr)
        print(json.dumps(compact_line(line), separators=(",", ":")))
    ctx.close()
    ranks.close()


if __name__ == "__main__":
    main()
