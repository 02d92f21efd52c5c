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
            dist.init_process_group(backend, **kw)
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

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
