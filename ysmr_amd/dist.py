"""One process per GPU, independent video streams, no data-path collective.

The reference parallelises over files with one worker process each (ysmr/main.py:281-288); here
each rank owns one GPU and a disjoint shard of the streams.  ``torch.distributed`` (RCCL on GPUs,
gloo on CPU for tests) is used only for the start/stop barrier and the max-over-ranks wall time of
the benchmark -- never for detections, tracks or rows.
"""
from __future__ import annotations

import os

__all__ = ["RankInfo", "rank_info", "init", "shard", "barrier", "max_over_ranks", "sum_over_ranks", "finish", "pci_bus_id", "local_cpus",
           "pin_to_gpu", "physical_device", "usable_gpus"]


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


def sum_over_ranks(value: float, info: RankInfo, device="cpu") -> float:
    """All-reduced sum (bench.py: how many ranks finished their steps, how many frames they linked)."""
    if info.world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value: float, info: RankInfo, device="cpu") -> list:
    """Every rank's value, in rank order, on every rank (bench.py: each rank's own frames/s, so that a SCALE record shows a
    straggler rank and not only the max-over-ranks time).  One all-reduce of a vector that is zero but for the own entry."""
    if info.world <= 1:
        return [float(value)]
    import torch
    import torch.distributed as dist
    t = torch.zeros(info.world, dtype=torch.float64, device=device)
    t[info.rank] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def finish(info: RankInfo):
    if info.world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


# ---- placement of a per-GPU worker on the host (ysmr(multiprocess=True), bench.py --gpus N) ---------------------------
# A rank / worker feeds ONE GPU a launch every ~11 us from a handful of threads (the caller's, a feed thread, four readers,
# an upload stream).  On an 8-GPU node that is ~50 busy host threads; left to the scheduler they end up on the far socket
# and share cores, and the link chain -- which has only ~1.4x the host time it needs -- slows down.
def physical_device(ordinal, env=None) -> str:
    """The entry of the parent's HIP_VISIBLE_DEVICES (or CUDA_VISIBLE_DEVICES, which HIP honours too) that its device ``ordinal`` is; the ordinal
    itself when neither is set.  A worker process that is to see only that GPU gets this value as HIP_VISIBLE_DEVICES."""
    env = os.environ if env is None else env
    for key in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        listed = [v.strip() for v in env.get(key, "").split(",") if v.strip()]
        if listed:
            return listed[ordinal % len(listed)]
    return str(ordinal)


def usable_gpus(dev_root="/dev/dri"):
    """How many GPUs this process may open, WITHOUT initialising the HIP runtime: the render nodes it can read and write
    (a container's device cgroup hides or denies the others, while management libraries and sysfs still list them)."""
    try:
        nodes = [n for n in os.listdir(dev_root) if n.startswith("renderD")]
    except OSError:
        return 0
    return sum(os.access(os.path.join(dev_root, n), os.R_OK | os.W_OK) for n in nodes)


def pci_bus_id(index=0):
    """'0000:c1:00.0' of HIP device ``index`` of this process, or None.  Asked of the HIP runtime that torch has loaded
    (the library mapped into this process, found in /proc/self/maps): dlopen-ing "libamdhip64.so" by name may bring in a
    SECOND runtime -- a pip torch wheel carries its own copy -- and two runtimes on one GPU is one too many."""
    import ctypes
    try:
        import torch  # noqa: F401  (loads the runtime)
        path = None
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            return None
        hip = ctypes.CDLL(path)
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) != 0:
            return None
        return buf.value.decode().lower()
    except (OSError, ImportError):
        return None


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def local_cpus(bus_id, sysfs_root="/sys"):
    """The CPUs of the NUMA node the PCI device ``bus_id`` hangs off (sysfs ``local_cpulist``; the node's cpulist when
    only ``numa_node`` is there); None when sysfs does not say (a VM, numa_node = -1)."""
    if not bus_id:
        return None
    dev = os.path.join(sysfs_root, "bus", "pci", "devices", bus_id)
    try:
        with open(os.path.join(dev, "local_cpulist")) as fh:
            cpus = _parse_cpulist(fh.read())
        if cpus:
            return cpus
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(dev, "numa_node")) as fh:
            node = int(fh.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs_root, "devices", "system", "node", "node{}".format(node), "cpulist")) as fh:
            return _parse_cpulist(fh.read()) or None
    except (OSError, ValueError):
        return None


def pin_to_gpu(index=0, sysfs_root="/sys", bus_id=None, allowed=None, setter=None, min_cpus=4):
    """Restrict this process to the CPUs next to GPU ``index`` (those of them it is allowed to use at all).  Returns the
    CPU set it pinned to, or None when nothing is known or fewer than ``min_cpus`` would be left (a worker runs about six
    busy threads).  ``bus_id`` / ``allowed`` / ``setter`` stand in for the HIP runtime and the scheduler calls in tests."""
    cpus = local_cpus(bus_id if bus_id is not None else pci_bus_id(index), sysfs_root)
    if not cpus:
        return None
    try:
        allowed = os.sched_getaffinity(0) if allowed is None else set(allowed)
    except (AttributeError, OSError):
        return None
    cpus &= allowed
    if len(cpus) < max(1, min_cpus):
        return None
    try:
        (setter or (lambda c: os.sched_setaffinity(0, c)))(cpus)
    except OSError:
        return None
    return cpus
