"""N>1 control path on CPU (gloo, world size 2): streams shard disjointly and completely, ranks do not
talk on the data path, rank 0 aggregates units / MAX-time exactly as bench.py does with RCCL."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def test_shard_streams_partitions():
    from mpeg_amd.shard import shard_streams
    for total, world in ((8192, 8), (8192, 1), (10, 4), (3, 8), (1024, 2)):
        got = [s for r in range(world) for s in shard_streams(total, world, r)]
        assert got == list(range(total))
    assert shard_streams(8192, 8, 3) == range(3072, 4096)   # BASELINE config 5: 1024 streams per GPU


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    import emu
    from mpeg_amd import synth
    from mpeg_amd.shard import Ranks, shard_streams
    from oracle import pyoracle
    ranks = Ranks(backend="gloo")
    mine = shard_streams(6, world, rank)                        # 6 independent streams over 2 ranks
    w, h = 64, 48
    seqs = {s: synth.generate_sequence(w, h, 3, seed=1000 + s) for s in mine}
    stores = {s: emu.EmuStore(w, h) for s in mine}     # stand-in for this rank's device (test-only emulator)
    units = [0]

    def body():
        for s in mine:
            for sub in seqs[s]:
                stores[s].submit(sub.pics, sub.mbs, sub.coefs)
                units[0] += len(sub.mbs)

    elapsed = ranks.timed(body)
    total_units = ranks.sum(units[0])
    ok = True
    for s in mine:                                              # every rank checks its own shard against the oracle
        ref = pyoracle.OracleStore(w, h)
        for sub in seqs[s]:
            ref.submit(sub.pics, sub.mbs, sub.coefs)
        for slot in range(3):
            ok &= all(np.array_equal(a, b) for a, b in zip(ref.read_planes(0, slot), stores[s].read_planes(0, slot)))
    all_ok = ranks.sum(1.0 if ok else 0.0)
    rates = ranks.gather(100.0 + rank)                          # per-rank values in rank order, as bench.py's per_rank_value
    assert rates == [100.0, 101.0] and 0 < ranks.last_local <= elapsed
    out.put((rank, list(mine), elapsed, total_units, all_ok))
    ranks.close()


def test_two_ranks_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, e0, u0, k0), (r1, s1, e1, u1, k1) = res
    assert s0 == [0, 1, 2] and s1 == [3, 4, 5]                  # disjoint, complete
    assert e0 == e1 and e0 > 0                                  # both ranks hold the MAX elapsed
    assert u0 == u1 == 6 * 3 * 12                               # 6 streams x 3 pictures x 12 macroblocks, summed over ranks
    assert k0 == k1 == 2.0                                      # both shards bit-exact vs the oracle


def test_ranks_that_share_a_gpu_are_refused_unless_asked_for():
    """bench.py's N > 1 line: n_gpus = the DISTINCT physical devices of the ranks (PCI addresses), and a launch whose ranks
    share a device exits non-zero unless --share-devices says it is meant (round-4 review: `local_rank %= n_dev` let an
    "8-GPU" number be measured on fewer).  mpeg_amd.shard.device_census is the rule, bench.py only gathers the addresses."""
    from mpeg_amd.shard import device_census
    eight = ["0000:%02x:00.0" % (0x05 + 0x10 * i) for i in range(8)]
    c = device_census(eight)
    assert c == {"devices": eight, "n_gpus": 8, "ranks": 8, "shared": False}
    with pytest.raises(ValueError, match="--share-devices"):
        device_census(eight[:4] * 2)                                  # 8 ranks on 4 GPUs: refused
    with pytest.raises(ValueError):
        device_census(["0000:05:00.0"] * 2)
    c = device_census(["0000:05:00.0"] * 8, share_devices=True)       # the functional run on a one-GPU box
    assert c["n_gpus"] == 1 and c["ranks"] == 8 and c["shared"] is True
    assert device_census(["0000:05:00.0"]) == {"devices": ["0000:05:00.0"], "n_gpus": 1, "ranks": 1, "shared": False}
    # bench.py applies it before any leg runs, and its lines carry the census, not the world size
    src = (ROOT / "bench.py").read_text()
    assert "device_census(ranks.gather_object(ctx.pci_bus_id()), args.share_devices)" in src
    assert '"n_gpus": census["n_gpus"], "ranks": world' in src and '"n_gpus": world' not in src and '"n_gpus": ranks.world' not in src
    assert '"untimed_prewarm_steps"' in src and '"effective_cores"' in src


def _census_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path[:0] = [str(ROOT)]
    from mpeg_amd.shard import Ranks, device_census
    ranks = Ranks(backend="gloo")
    ids = ranks.gather_object("0000:05:00.0" if rank < 2 else "0000:15:00.0")   # ranks 0 and 1 sit on one device
    try:
        device_census(ids)
        refused = False
    except ValueError:
        refused = True
    out.put((rank, ids, refused, device_census(ids, share_devices=True)["n_gpus"]))
    ranks.close()


def test_device_census_over_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_census_worker, args=(r, 3, port, out)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ids, refused, n in res:       # every rank sees the same census and takes the same decision
        assert ids == ["0000:05:00.0", "0000:05:00.0", "0000:15:00.0"] and refused and n == 2


def test_effective_cores_reads_the_cgroup_quota(tmp_path, monkeypatch):
    from mpeg_amd import shard
    e = shard.effective_cores()
    assert e["affinity_cpus"] >= 1 and 0 < e["effective_cores"] <= e["affinity_cpus"]
    assert e["cgroup_quota_cores"] is None or e["effective_cores"] == min(e["affinity_cpus"], e["cgroup_quota_cores"])
    # a v2 quota of 10.5 cores under a wide mask
    real_open = open
    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            f = tmp_path / "cpu.max"
            f.write_text("1050000 100000\n")
            return real_open(f, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr("builtins.open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    e = shard.effective_cores()
    assert e == {"affinity_cpus": 256, "cgroup_quota_cores": 10.5, "effective_cores": 10.5}


def test_numa_helpers_parse_sysfs_and_never_raise():
    """mpeg_amd.shard.pin_to_node: the rank's process goes to the cores of its GPU's NUMA node; unknown nodes, or a
    platform without sysfs / sched_setaffinity, change nothing."""
    import os
    from mpeg_amd import shard
    assert shard.pin_to_node(-1) == 0 and shard.pin_to_node(None) == 0 and shard.pin_to_node(4096) == 0
    node0 = shard.cpus_of_node(0)
    if node0 and hasattr(os, "sched_getaffinity"):
        before = os.sched_getaffinity(0)
        try:
            n = shard.pin_to_node(0)
            assert n == len(node0 & before) and (n == 0 or os.sched_getaffinity(0) == (node0 & before))
        finally:
            os.sched_setaffinity(0, before)
