"""mpeg::VideoBatch (mpeg_amd/host/batch.cpp): many streams parsed on the CPU, ONE reconstruction call per
tick.  CPU only: the store is the test-only lane emulator; the HIP store runs the same cases in
test_gpu_golden.py.  Every stream must come out exactly as if it had been decoded alone — the golden
hash of the reference's TestVideoGolden (mpeg_test.go:205-231) — whatever the others are doing."""
import numpy as np
import pytest

import hostlib

VIDEO_HASH = 0xea6d7fcb1340ba3f       # testdata/test.mpeg1video, damaged stream (53 invalid blocks, duplicates)
TESTMPG_VIDEO_HASH = 0xd00818edcafdc702


def run_batch(oracle, streams, delays, device=None, threads=1, device_pack=True):
    """streams[i] joins the batch after delays[i] ticks; returns per-stream (hash, frames) + counters."""
    b = hostlib.HostBatch(len(streams), device=device, threads=threads)
    b.set_device_pack(device_pack)
    h = [oracle.FNV_OFFSET] * len(streams)
    n = [0] * len(streams)
    added, tick = 0, 0
    order = sorted(range(len(streams)), key=lambda i: delays[i])
    index_of = {}
    while True:
        while added < len(order) and delays[order[added]] <= tick:
            index_of[b.add_stream(streams[order[added]])] = order[added]
            added += 1
        produced = b.decode_all()
        for k, i in index_of.items():
            f = b.frame(k)
            if f is not None:
                for p in hostlib.frame_planes(f):
                    h[i] = oracle.fnv1a64(p, h[i])
                n[i] += 1
        tick += 1
        if produced == 0 and added == len(order):
            break
    c = b.counters()
    b.close()
    return h, n, c


def test_lockstep_streams_share_one_device_call_per_tick(oracle, golden_dir):
    es = (golden_dir / "test.mpeg1video").read_bytes()
    h, n, c = run_batch(oracle, [es] * 6, [0] * 6)
    assert h == [VIDEO_HASH] * 6 and n == [260] * 6
    # 6 streams x ~274 pictures, but about one device call per tick (extra ones: the first reference picture
    # of a stream yields no frame -> a second round in that tick; duplicated macroblock addresses re-submit)
    assert c["queued_pictures"] >= 6 * 260 and c["device_submits"] < 1.5 * 260


def test_staggered_and_mixed_streams(oracle, golden_dir):
    """Streams at different positions of their GOPs in every call (I, P and B pictures of different
    streams side by side in one submit), one of them a different bitstream of the same picture size."""
    es = (golden_dir / "test.mpeg1video").read_bytes()
    clean = oracle.ps_extract((golden_dir / "test.mpg").read_bytes(), 0xE0)[0]
    streams = [es, clean, es, es, clean]
    h, n, c = run_batch(oracle, streams, [0, 0, 1, 5, 9])
    assert h == [VIDEO_HASH, TESTMPG_VIDEO_HASH, VIDEO_HASH, VIDEO_HASH, TESTMPG_VIDEO_HASH]
    assert n == [260, 278, 260, 260, 278]


@pytest.mark.parametrize("threads", [2, 5])
def test_threaded_parse_gives_the_same_frames_and_device_calls(oracle, golden_dir, threads):
    """VideoBatch::SetThreads: streams parsed on a pool, their device requests recorded and replayed in stream
    order — same frames; no more device calls than with one thread (replaying the k-th request of every
    stream together groups re-submits of damaged streams better than stream-after-stream does)."""
    es = (golden_dir / "test.mpeg1video").read_bytes()
    clean = oracle.ps_extract((golden_dir / "test.mpg").read_bytes(), 0xE0)[0]
    streams = [es, clean, es, es, clean, es, es]
    delays = [0, 0, 1, 5, 9, 0, 2]
    h1, n1, c1 = run_batch(oracle, streams, delays)
    h, n, c = run_batch(oracle, streams, delays, threads=threads)
    want = {id(es): (VIDEO_HASH, 260), id(clean): (TESTMPG_VIDEO_HASH, 278)}
    assert h == [want[id(s)][0] for s in streams] and n == [want[id(s)][1] for s in streams]
    assert (h, n) == (h1, n1) and c["queued_pictures"] == c1["queued_pictures"]
    assert c["device_submits"] <= c1["device_submits"]
    # and the pictures went through staged submits (put from the pool), not through the merged path
    assert c1["staged_commits"] == 0 and c["staged_commits"] > 0.8 * c["device_submits"]
    # ... packed by the DEVICE packer (here: its lane functions) — the damaged golden stream's snapshot blocks, invalid intra
    # blocks and re-submits included; with the packing left to the host: the same frames
    assert c["device_pack_stages"] == c["staged_commits"]
    h0, n0, c0 = run_batch(oracle, streams, delays, threads=threads, device_pack=False)
    assert (h0, n0) == (h, n) and c0["device_pack_stages"] == 0 and c0["staged_commits"] == c["staged_commits"]


def test_different_picture_sizes_are_refused(oracle, golden_dir):
    es = (golden_dir / "test.mpeg1video").read_bytes()
    # same stream with the 12-bit horizontal size in the sequence header patched from 160 to 176
    i = es.find(b"\x00\x00\x01\xb3")
    assert i >= 0 and (es[i + 4] << 4 | es[i + 5] >> 4) == 160
    other = bytearray(es)
    other[i + 4], other[i + 5] = 176 >> 4, ((176 & 15) << 4) | (es[i + 5] & 15)
    b = hostlib.HostBatch(2)
    b.add_stream(es)
    with pytest.raises(RuntimeError, match="same picture size"):
        b.add_stream(bytes(other))
    b.close()


# ------------------------------------------------------------------------------------------------ audio
AUDIO_HASH = 0xf1b76cdf8e6cdea5   # TestAudioGolden, no FMA (mpeg_test.go:193-197)


def run_audio_batch(oracle, n, delays, fmt=0, device=None, window=None, threads=1):
    """n copies of test.mp2; stream i joins after delays[i] ticks.  Returns per-stream (hash, frames), device calls.
    threads > 1: the streams' frames of a tick are parsed on that many host threads (AudioBatch::SetThreads)."""
    from pathlib import Path
    mp2 = (Path(__file__).resolve().parent / "golden" / "test.mp2").read_bytes()
    b = hostlib.HostAudioBatch(n, device=device, fmt=fmt, window=window)
    if threads > 1:
        hostlib.host().mpeghost_audio_batch_set_threads(b.h, threads)
    h, cnt = [oracle.FNV_OFFSET] * n, [0] * n
    added, tick = 0, 0
    order = sorted(range(n), key=lambda i: delays[i])
    index_of = {}
    while True:
        while added < n and delays[order[added]] <= tick:
            index_of[b.add_stream(mp2)] = order[added]
            added += 1
        produced = b.decode_all()
        for k, i in index_of.items():
            s = b.samples(k)
            if s is not None:
                h[i] = oracle.fnv1a64(s, h[i])
                cnt[i] += 1
        tick += 1
        if produced == 0 and added == n:
            break
    calls = b.device_calls
    b.close()
    return h, cnt, calls


@pytest.mark.parametrize("threads", [1, 3])
def test_audio_batch_streams_share_one_synthesis_call_per_tick(oracle, emu, threads):
    win = (np.array(emu._window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)
    h, cnt, calls = run_audio_batch(oracle, 5, [0, 0, 3, 7, 40], window=win, threads=threads)
    assert h == [AUDIO_HASH] * 5 and cnt == [355] * 5      # every stream exactly as if decoded alone
    assert calls == 355 + 40                                # one call per tick while any stream is alive


def test_streams_sharded_over_two_stores(oracle, golden_dir):
    """mpeg::ShardedVideoBatch over two (emulator) stores: stream s lives on store s mod 2, each shard is driven by its
    own host thread with its own device calls, every stream still comes out as if decoded alone, and no shard sees
    another one's pictures (no collective)."""
    es = (golden_dir / "test.mpeg1video").read_bytes()
    clean = oracle.ps_extract((golden_dir / "test.mpg").read_bytes(), 0xE0)[0]
    streams = [es, clean, es, clean, es]
    b = hostlib.HostSharded(len(streams), 2)
    b.set_threads(3)      # every shard parses on its own pool (pinned to its GPU's NUMA node when there is a GPU)
    for s in streams:
        b.add_stream(s)
    assert [b.device_of(i) for i in range(5)] == [0, 1, 0, 1, 0]
    h, n = [oracle.FNV_OFFSET] * 5, [0] * 5
    while b.decode_all():
        for i in range(5):
            f = b.frame(i)
            if f is not None:
                for p in hostlib.frame_planes(f):
                    h[i] = oracle.fnv1a64(p, h[i])
                n[i] += 1
    assert h == [VIDEO_HASH, TESTMPG_VIDEO_HASH, VIDEO_HASH, TESTMPG_VIDEO_HASH, VIDEO_HASH]
    assert n == [260, 278, 260, 278, 260]
    c0, c1 = b.counters(0), b.counters(1)
    assert c0["queued_pictures"] >= 3 * 260 and c1["queued_pictures"] >= 2 * 278   # each shard queued its own streams only
    assert c0["queued_pictures"] + c1["queued_pictures"] < 5 * 300
    b.close()


def test_sharded_batch_passes_device_pack_and_sync_through(oracle, golden_dir):
    """ShardedVideoBatch::SetDevicePack / Sync reach every shard (round-4 advisor: the sharded driver had neither, so a
    device-packed commit's deferred verdict could not be asked for): the same streams, device-packed on both shards, with a
    Sync after every tick — same frames."""
    es = (golden_dir / "test.mpeg1video").read_bytes()
    b = hostlib.HostSharded(3, 2)
    b.set_threads(2)
    b.set_device_pack(True)
    for _ in range(3):
        b.add_stream(es)
    h, n = [oracle.FNV_OFFSET] * 3, [0] * 3
    while b.decode_all():
        b.sync()
        for i in range(3):
            f = b.frame(i)
            if f is not None:
                for p in hostlib.frame_planes(f):
                    h[i] = oracle.fnv1a64(p, h[i])
                n[i] += 1
    b.sync()
    assert h == [VIDEO_HASH] * 3 and n == [260] * 3
    b.close()


def test_pools_are_sized_by_the_cpu_time_the_process_gets():
    """SetThreads is a request: a pool never has more threads than the process has CPU time for (affinity mask capped by the
    cgroup quota, rounded up) — BENCH_r04: 64 threads under a 10-core quota parsed 26 % slower than 16.  0 = as many as fit."""
    import math
    import os
    L = hostlib.host()
    eff = L.mpeghost_effective_cores()
    assert 1 <= eff <= len(os.sched_getaffinity(0))
    from mpeg_amd.shard import effective_cores
    assert abs(effective_cores()["effective_cores"] - eff) < 1e-6        # bench.py's figure is the library's
    cap = max(1, math.ceil(eff))
    b = hostlib.HostBatch(4, threads=1)
    for asked, got in ((1, 1), (2, min(2, cap)), (4096, cap), (0, cap)):
        L.mpeghost_batch_set_threads(b.h, asked)
        assert L.mpeghost_batch_threads(b.h) == got, (asked, got)
    b.close()


def test_audio_batch_with_a_pool_flushes_a_frame_decoded_outside_the_batch_tick(oracle, emu):
    """A stream of the batch decoded DIRECTLY between two ticks leaves its slot occupied; the pooled DecodeAll used to throw
    std::logic_error where the one-thread path flushes and goes on (round-4 advisor).  Both now flush first."""
    from pathlib import Path
    mp2 = (Path(__file__).resolve().parent / "golden" / "test.mp2").read_bytes()
    win = (np.array(emu._window_x2(), np.float32) * np.float32(0.5)).astype(np.float32)
    got = {}
    for threads in (1, 3):
        b = hostlib.HostAudioBatch(3, window=win)
        hostlib.host().mpeghost_audio_batch_set_threads(b.h, threads)
        for _ in range(3):
            b.add_stream(mp2)
        h = [oracle.FNV_OFFSET] * 3
        for tick in range(12):
            if tick == 5:
                assert b.decode_stream_directly(1)          # stream 1 runs one frame ahead, outside the tick
            assert b.decode_all() == 3
            for k in range(3):
                h[k] = oracle.fnv1a64(b.samples(k), h[k])
        got[threads] = h
        b.close()
    assert got[1] == got[3] and got[1][0] == got[1][2] != got[1][1]


def _cgroup_tree(tmp_path, files, proc_text):
    root = tmp_path / "cg"
    for rel, text in files.items():
        f = root / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(text)
    root.mkdir(exist_ok=True)
    proc = tmp_path / "proc_self_cgroup"
    proc.write_text(proc_text)
    return str(root).encode(), str(proc).encode()


@pytest.mark.parametrize("files,proc,want", [
    ({"cpu.max": "1000000 100000\n"}, "0::/\n", 10.0),                                              # the mount's root (a container with a cgroup namespace)
    ({"cpu.max": "max 100000\n", "kube/pod1/cpu.max": "400000 100000\n", "kube/pod1/ctr/cpu.max": "max 100000\n"},
     "0::/kube/pod1/ctr\n", 4.0),                                                                    # nested, the quota on the PARENT (round-5 advisor)
    ({"cpu.max": "max 100000\n", "a/cpu.max": "800000 100000\n", "a/b/cpu.max": "250000 100000\n"}, "0::/a/b\n", 2.5),   # the tightest of the chain
    ({"cpu.max": "max 100000\n", "a/cpu.max": "200000 100000\n"}, "0::/other\n", 0.0),               # not our branch
    ({"cpu/cpu.cfs_quota_us": "-1\n", "cpu/cpu.cfs_period_us": "100000\n", "cpu/docker/x/cpu.cfs_quota_us": "350000\n",
      "cpu/docker/x/cpu.cfs_period_us": "100000\n"}, "12:pids:/docker/x\n5:cpu,cpuacct:/docker/x\n", 3.5),               # v1, nested
    ({"cpu/cpu.cfs_quota_us": "600000\n", "cpu/cpu.cfs_period_us": "100000\n"}, "5:cpuacct,cpu:/\n", 6.0),               # v1, root
    ({}, "0::/nowhere\n", 0.0),
])
def test_the_cgroup_quota_is_the_tightest_of_the_process_and_its_ancestors(tmp_path, files, proc, want):
    """mpeg::CgroupQuotaCores and its Python restatement (bench.py's effective_cores) on made-up hierarchies: the process's cgroup
    from /proc/self/cgroup, every ancestor up to the mount's root, v2 and v1."""
    from mpeg_amd.shard import cgroup_quota_cores
    root, procf = _cgroup_tree(tmp_path, files, proc)
    got = hostlib.host().mpeghost_cgroup_quota_cores(root, procf)
    assert abs(got - want) < 1e-9, got
    assert abs(cgroup_quota_cores(root.decode(), procf.decode()) - want) < 1e-9


def test_sharded_pools_share_one_thread_budget():
    """ShardedVideoBatch::SetThreads(n): n threads for ALL shards together, never more than the process has CPU time for, at least
    one per shard (round-5 advisor: n went to EVERY shard — G x the quota on a G-GPU node)."""
    import math
    L = hostlib.host()
    cap = max(1, math.ceil(L.mpeghost_effective_cores()))
    for shards in (2, 3):
        b = hostlib.HostSharded(6, shards)
        for asked in (0, 1, shards, 4096):
            b.set_threads(asked)
            total = L.mpeghost_sharded_threads(b.h)
            want = max(shards, cap if asked == 0 else min(asked, cap))
            assert total == want, (shards, asked, total, want)
        b.close()


@pytest.mark.parametrize("threads,fetch_around", [(2, True), (4, True), (3, False)])
def test_a_refused_picture_is_reported_by_the_next_call_and_costs_the_other_streams_nothing(oracle, golden_dir, threads, fetch_around):
    """Device-packed hand-over is the default (round 6), and its error contract is the reference's unit of failure — the picture:
    stream 2's picture of tick k arrives damaged (test hook: a quantiser scale of 0, what no parser emits and every validator
    refuses).  Tick k itself returns normally (the verdict is deferred); the NEXT DecodeAll throws — RefusedStreams() = [2] —
    before anything of its own round reaches the device, and the call after that goes on: with fetch the refusal is known when
    tick k's frames are read back and thrown before tick k + 1 parses; without, tick k + 1's round is parsed, HELD, and committed
    by the next call.  Either way nobody loses a frame, and every other stream's frames, all the way to the end, are the golden
    ones: the commit that carried the damaged picture reconstructed theirs."""
    run_refusal(oracle, (golden_dir / "test.mpeg1video").read_bytes(), None, threads, fetch_around)


def run_refusal(oracle, es, device, threads, fetch_around=True):
    n_streams, victim, at_tick = 5, 2, 20
    b = hostlib.HostBatch(n_streams, device=device, threads=threads)
    assert b.device_pack                                       # the default
    for _ in range(n_streams):
        b.add_stream(es)
    # per stream the hash of every frame: the ticks around the damage may run without fetch (their frames stay on the device)
    frames = [[] for _ in range(n_streams)]
    tick, reported_at = 0, None
    while True:
        if tick == at_tick:
            b.damage_next_picture(victim)
        fetch = fetch_around or not (at_tick - 2 <= tick <= at_tick + 4)
        try:
            produced = b.decode_all(fetch)
        except RuntimeError as e:
            assert reported_at is None and tick == at_tick + 1, (tick, str(e))      # the call right after
            assert "refused" in str(e) and b.refused_streams() == [victim]
            reported_at = tick
            tick += 1
            continue
        for i in range(n_streams):
            f = b.frame(i)
            if f is not None:
                frames[i].append(hash(np.concatenate(hostlib.frame_planes(f)).tobytes()) if fetch else None)
        tick += 1
        if produced == 0:
            break
    b.sync()
    b.close()
    assert reported_at == at_tick + 1
    assert [len(x) for x in frames] == [260] * n_streams                            # nobody lost a frame
    clean = hostlib.HostBatch(1, device=device, threads=1)
    clean.add_stream(es)
    want = []
    while clean.decode_all():
        want.append(hash(np.concatenate(hostlib.frame_planes(clean.frame(0))).tobytes()))
    clean.close()
    for i in range(n_streams):
        same = [a == w for a, w in zip(frames[i], want) if a is not None]
        if i == victim:
            assert not all(same)                                                     # (its picture really was dropped)
        else:
            assert all(same) and len(same) >= 250, i                                 # the golden frames, before and after
