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
