"""Multi-GPU control path: independent streams sharded over ranks, no data-path collective.

Streams share nothing: rank r decodes its own streams into its own frame store.
torch.distributed is used only for the barrier around the timed region and for reducing the
timings / unit counts to rank 0.  The backend is "gloo" everywhere — the north star asks for no RCCL, and a
control plane of three scalars per run has no use for it ("nccl" still works if a caller asks for it)."""
from __future__ import annotations

import os
import time


def shard_streams(total_streams: int, world: int, rank: int) -> range:
    """Contiguous shard of `total_streams` for `rank` (8192 streams over 8 GPUs -> 1024 each)."""
    base, extra = divmod(total_streams, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def cpus_of_node(node: int):
    """CPUs of a host NUMA node (sysfs cpulist), or None."""
    try:
        txt = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    except OSError:
        return None
    cpus = set()
    for part in txt.split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus or None


def pin_to_node(node: int):
    """Bind this process (and the host threads it starts: they inherit the mask) to the cores of the NUMA node its GPU is
    attached to — one process per GPU on a two-socket node, each feeding its device from its own socket.  Returns the
    number of CPUs bound to, or 0 if nothing was changed (unknown node, no permission)."""
    cpus = cpus_of_node(node) if node is not None and node >= 0 else None
    if not cpus:
        return 0
    try:
        allowed = os.sched_getaffinity(0)
        want = cpus & allowed
        if not want:
            return 0
        os.sched_setaffinity(0, want)
        return len(want)
    except (AttributeError, OSError):
        return 0


def _quota_of_cgroup_dir(path: str, v2: bool):
    """One cgroup directory's CPU-time quota in cores, or 0.0 (none there)."""
    try:
        if v2:
            txt = open(os.path.join(path, "cpu.max")).read().split()
            return float(txt[0]) / float(txt[1]) if len(txt) == 2 and txt[0] != "max" and float(txt[1]) > 0 else 0.0
        q = float(open(os.path.join(path, "cpu.cfs_quota_us")).read())
        per = float(open(os.path.join(path, "cpu.cfs_period_us")).read())
        return q / per if q > 0 and per > 0 else 0.0
    except (OSError, ValueError, IndexError):
        return 0.0


def cgroup_quota_cores(root: str = "/sys/fs/cgroup", proc_file: str = "/proc/self/cgroup") -> float:
    """The tightest CPU-time quota (cores; 0.0 = none) that applies to this process: its own cgroup's and every ancestor's, v2
    (cpu.max) and v1 (cpu/cpu.cfs_*), found through /proc/self/cgroup — a process in a nested cgroup (no cgroup namespace: a
    systemd slice, a pod without cgroupns) never sees its quota in the mount's root files.  The library's rule
    (mpeg::CgroupQuotaCores, batch.cpp), restated: tests compare the two on made-up hierarchies."""
    v2_path = v1_path = None
    try:
        for line in open(proc_file).read().splitlines():
            parts = line.split(":", 2)
            if len(parts) != 3:
                continue
            if parts[0] == "0" and parts[1] == "":
                v2_path = parts[2]
            elif "cpu" in parts[1].split(","):
                v1_path = parts[2]
    except OSError:
        pass
    tightest = 0.0

    def walk(mount, path, v2):
        nonlocal tightest
        while True:
            q = _quota_of_cgroup_dir(mount + ("" if path == "/" else path), v2)
            if q > 0 and (tightest == 0 or q < tightest):
                tightest = q
            if path in ("", "/"):
                return
            cut = path.rfind("/")
            path = "/" if cut <= 0 else path[:cut]

    walk(root, v2_path if v2_path is not None else "/", True)
    walk(os.path.join(root, "cpu"), v1_path if v1_path is not None else "/", False)
    return tightest


def effective_cores():
    """The CPU time this process can really get, in cores: the cgroup quota that applies to it (cgroup_quota_cores) if one is set,
    capped by the affinity mask; None for the quota part if there is none.  A container that exposes 256 hardware threads under a
    10-core quota runs 256 threads at 10 cores' worth of time: pools are sized, and baselines quoted, by this."""
    mask = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cgroup_quota_cores() or None
    return {"affinity_cpus": mask, "cgroup_quota_cores": quota, "effective_cores": min(float(mask), quota) if quota else float(mask)}


def device_census(identities, share_devices: bool = False):
    """identities: every rank's physical-device identity (PCI address), in rank order.  One process per GPU is the contract;
    ranks that share a physical device are refused unless asked for (`--share-devices`: a functional run of the N > 1 path on a
    smaller box), and then the line says how many devices it was really measured on.  Returns {"devices", "n_gpus", "ranks",
    "shared"}; raises ValueError for the refusal."""
    distinct = sorted(set(identities))
    out = {"devices": list(identities), "n_gpus": len(distinct), "ranks": len(identities), "shared": len(distinct) < len(identities)}
    if out["shared"] and not share_devices:
        raise ValueError("%d ranks on %d distinct GPUs (%s): one process per GPU is the contract; pass --share-devices for a "
                         "functional run of the multi-rank path on fewer devices (the line then reports n_gpus = %d)" %
                         (len(identities), len(distinct), ", ".join(distinct), len(distinct)))
    return out


class Ranks:
    """RANK / LOCAL_RANK / WORLD_SIZE from the launcher's environment; process group only if world > 1."""

    def __init__(self, backend: str = "gloo", device_id=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            kw = {"device_id": device_id} if (device_id is not None and backend == "nccl") else {}
            # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to 7 peer ranks ..."), where bench.py's
            # one JSON line belongs: stdout points at stderr while the group forms (the first barrier completes the mesh)
            import sys
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend, **kw)
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
            self.dist = dist
        self.backend = backend
        self.last_local = 0.0   # this rank's own elapsed time of the last timed() body

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def _tensor(self, value):
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        return torch.tensor([value], dtype=torch.float64, device=dev)

    def max(self, value: float) -> float:
        if self.dist is None:
            return value
        t = self._tensor(value)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value: float) -> float:
        if self.dist is None:
            return value
        t = self._tensor(value)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def timed(self, body, device_sync=lambda: None) -> float:
        """barrier + device sync on both sides of `body`; returns the MAX elapsed over ranks."""
        device_sync()
        self.barrier()
        t0 = time.perf_counter()
        body()
        device_sync()
        self.last_local = time.perf_counter() - t0
        self.barrier()
        return self.max(time.perf_counter() - t0)

    def gather(self, value: float) -> list:
        """Every rank's `value`, in rank order (on every rank)."""
        if self.dist is None:
            return [value]
        import torch
        mine = self._tensor(value)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [float(t.item()) for t in out]

    def gather_object(self, obj) -> list:
        """Every rank's (picklable) `obj`, in rank order (on every rank)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
