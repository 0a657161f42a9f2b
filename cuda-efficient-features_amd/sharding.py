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
