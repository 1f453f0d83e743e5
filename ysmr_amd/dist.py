"""One process per GPU, independent video streams, no data-path collective.

The reference parallelises over files with one worker process each (ysmr/main.py:281-288); here
each rank owns one GPU and a disjoint shard of the streams.  ``torch.distributed`` (RCCL on GPUs,
gloo on CPU for tests) is used only for the start/stop barrier and the max-over-ranks wall time of
the benchmark -- never for detections, tracks or rows.
"""
from __future__ import annotations

import os

__all__ = ["RankInfo", "rank_info", "init", "shard", "barrier", "max_over_ranks", "finish"]


class RankInfo:
    def __init__(self, rank, local_rank, world):
        self.rank, self.local_rank, self.world = rank, local_rank, world


def rank_info(env=None) -> RankInfo:
    env = os.environ if env is None else env
    return RankInfo(int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1")))


def init(info: RankInfo, backend="nccl", device=None):
    """Join the process group (no-op for a single process)."""
    if info.world <= 1:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        kwargs = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=info.rank, world_size=info.world, **kwargs)
    return dist


def shard(items, rank, world):
    """Stream i goes to rank i mod world (the reference hands path i to the next free worker)."""
    return [it for i, it in enumerate(items) if i % world == rank]


def barrier(info: RankInfo):
    if info.world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value: float, info: RankInfo, device="cpu") -> float:
    if info.world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def finish(info: RankInfo):
    if info.world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
