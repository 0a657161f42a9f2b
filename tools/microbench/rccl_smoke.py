"""RCCL smoke on one GPU: process group of one rank, the counter reduction of bench.py (MAX of time, SUM of keypoints)."""
import os, sys, torch
sys.path.insert(0, ".")
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
import cef_loader
sharding = cef_loader.load_submodule("sharding")
print(sharding.reduce_counters(dist, "cuda", 1.25, 320000, 8))
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group(); print("rccl ok")
