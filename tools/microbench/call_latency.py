#!/usr/bin/env python3
"""One-call latency (sample_benchmark.cpp:39-52 protocol: async call + device synchronise) of detectAndCompute BAD512:
python tools/microbench/call_latency.py [fhd|4k|8k] [iters]   (knobs come from the environment: EFX_BLUR_FORK_MIN_PX, EFX_BLUR_FORK, ...)"""
import sys, time
sys.path.insert(0, ".")
import torch
import cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
size = sys.argv[1] if len(sys.argv) > 1 else "fhd"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows, cols = synth.SIZES[size]
img = torch.from_numpy(synth.synth_frame(rows, cols, seed=1000)).cuda()
det = EF.create(40000, dtype=EF.BAD_512)
kps = torch.zeros((5, 40000), dtype=torch.float32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
desc = torch.zeros((40000, 64), dtype=torch.uint8, device="cuda")
for _ in range(5):
    det.detectAndComputeAsync(img, kps, desc, cnt); torch.cuda.synchronize()
ts = []
for _ in range(iters):
    t0 = time.perf_counter(); det.detectAndComputeAsync(img, kps, desc, cnt); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"{size}: n={int(cnt.item())} mean {sum(ts) / len(ts) * 1e3:.4f} ms  median {ts[len(ts) // 2] * 1e3:.4f}  min {ts[0] * 1e3:.4f}")
