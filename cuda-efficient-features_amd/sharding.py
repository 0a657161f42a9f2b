"""Frame sharding for multi-GPU runs: frames are independent, so rank r of W processes frames
r*F .. r*F+F-1 of each step (weak scaling: F frames per GPU per step) with no data-path collective; the only
communication is the reduction of the throughput counters (RCCL on GPUs, gloo in the CPU tests)."""


def frames_for_rank(frames_per_rank, rank, world):
    """Global frame indices (seed offsets) handled by `rank` in one step."""
    if not (0 <= rank < world) or frames_per_rank < 0:
        raise ValueError("bad rank / world / frames_per_rank")
    return [rank * frames_per_rank + i for i in range(frames_per_rank)]


def reduce_counters(dist, device, elapsed_s, keypoints, frames):
    """MAX of the elapsed time and SUM of the keypoint / frame counters over all ranks.
    `dist` is torch.distributed (initialised) or None for a single process."""
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    c = torch.tensor([float(keypoints), float(frames)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c[0].item()), float(c[1].item())


def gather_per_rank(dist, device, value):
    """The ranks' own values of one scalar (elapsed time, keypoints ...), in rank order, on every rank: one all-gather of a
    double -- a counter collective like the reductions above, never frame data."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist is None:
        return [float(value)]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def cpus_for_rank(allowed, local_rank, local_world):
    """The slice of the process's allowed CPUs that rank `local_rank` of `local_world` ranks on this node pins itself to:
    contiguous blocks of the sorted CPU list (Linux numbers the cores of a NUMA node contiguously, and the GPUs of a node are
    enumerated in NUMA order on the 8-GPU MI300-class boards, so block r is on or next to GPU r's node).  Every rank gets at
    least one CPU; with fewer CPUs than ranks they are shared round-robin."""
    cpus = sorted(int(c) for c in allowed)
    if not cpus or local_world < 1 or not (0 <= local_rank < local_world):
        raise ValueError("bad CPU set / local rank / local world")
    if len(cpus) < local_world:
        return [cpus[local_rank % len(cpus)]]
    per, rem = divmod(len(cpus), local_world)
    start = local_rank * per + min(local_rank, rem)
    return cpus[start:start + per + (1 if local_rank < rem else 0)]


def pin_to_rank_cpus(local_rank, local_world):
    """Pins the calling process to its slice (host threads of 8 ranks otherwise migrate over all sockets: the per-call enqueue
    of ~55 us is the only host work on the hot path, and it is latency that the max-over-ranks timing would show).  Returns the
    CPU list, or None where the platform has no sched_setaffinity."""
    import os
    if not hasattr(os, "sched_setaffinity"):
        return None
    mine = cpus_for_rank(os.sched_getaffinity(0), local_rank, local_world)
    os.sched_setaffinity(0, mine)
    return mine
