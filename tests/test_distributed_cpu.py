"""world_size-2 gloo test of the multi-GPU path's host logic: frame sharding without overlap and the
counter reduction that bench.py performs over RCCL on the GPU box."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cef_loader


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    sharding = cef_loader.load_submodule("sharding")
    frames = sharding.frames_for_rank(8, rank, world)
    # pretend every frame k yields 1000 + k keypoints and rank r needs (r + 1) seconds
    kp = sum(1000 + k for k in frames)
    t, kps, nfr = sharding.reduce_counters(dist, "cpu", float(rank + 1), kp, len(frames))
    dist.barrier()
    out.put((rank, frames, t, kps, nfr))
    dist.destroy_process_group()


def test_sharding_and_counter_reduction_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_frames = sum((r[1] for r in res), [])
    assert sorted(all_frames) == list(range(16))                 # every frame exactly once
    for _, _, t, kps, nfr in res:
        assert t == 2.0                                          # MAX over ranks
        assert kps == sum(1000 + k for k in range(16)) and nfr == 16.0


def test_single_process_passthrough():
    sharding = cef_loader.load_submodule("sharding")
    assert sharding.frames_for_rank(3, 0, 1) == [0, 1, 2]
    assert sharding.reduce_counters(None, "cpu", 1.5, 10, 2) == (1.5, 10.0, 2.0)
