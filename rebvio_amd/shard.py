"""Multi-GPU sharding of the hot path: independent camera streams, one per rank/GPU, no data-path collective
(each rebvio::Rebvio instance owns its detector, tracker and queues - reference rebvio.hpp:91-112).
torch.distributed is used only for the bench's barrier and the max-over-ranks of the timed region."""
from __future__ import annotations

import os


def env_ranks():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def stream_id_for_rank(rank: int) -> int:
    """Rank r processes camera stream r (its own scene seed, frames, detector servo and tracker state)."""
    return rank


def _parse_cpulist(text: str) -> set:
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa_node(pci_bus_id: str, sysfs: str = "/sys") -> int:
    """Pins the calling thread (and every thread it creates afterwards: the detect worker, the pinned staging buffers'
    first touch) to the CPUs of the NUMA node the GPU hangs off, so that on a two-socket 8-GPU node no rank polls or fills
    its pinned glue slots across the socket link. pci_bus_id as "0000:c1:00.0". Returns the node, or -1 when nothing was
    changed (single-node host, node unknown, or none of its CPUs are in this process's allowed set)."""
    try:
        with open(f"{sysfs}/bus/pci/devices/{pci_bus_id.lower()}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return -1
        with open(f"{sysfs}/devices/system/node/node{node}/cpulist") as f:
            cpus = _parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0)
        target = cpus & allowed
        if len(target) < 2 or target == allowed:
            return -1 if len(target) < 2 else node
        os.sched_setaffinity(0, target)
        return node
    except (OSError, ValueError):
        return -1


def init_group(backend: str, rank: int, world: int, device=None):
    import torch.distributed as dist
    if world <= 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def max_over_ranks(seconds: float, world: int, device="cpu") -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, world: int, device="cpu") -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def whole_job_fps(world: int, steps_per_rank: int, tmax: float) -> float:
    """Every rank pushes `steps_per_rank` frames of its own stream: aggregate = all frames / slowest rank's time."""
    return world * steps_per_rank / tmax
