#!/usr/bin/env python3
"""Config C3 alone (compute-only BAD on exactly 40 000 keypoints of the 4K frame, tools/workloads.py), for rocprofv3:
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3 -o c3 -- python tools/microbench/c3_run.py [--iters 30] [--nbits 512]
Prints the host-clock time per call (async call + stream sync, sample_benchmark.cpp:39-52)."""
import argparse
import sys
import time

sys.path.insert(0, ".")
import torch

import cef_loader
from tools import workloads

cef = cef_loader.load()
EF = cef.EfficientFeatures
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--nbits", type=int, default=512)
args = ap.parse_args()
img = torch.from_numpy(workloads.frame_c34()).cuda()
kps = torch.zeros((5, 40000), dtype=torch.float32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
det = EF.create(workloads.N40K, 1.2, 8, 0, 20, workloads.C34_NMS_RADIUS, EF.BAD_256)
det.detectAsync(img, kps, cnt); torch.cuda.synchronize()
n = int(cnt.item())
d = EF.create(40000, dtype=EF.BAD_512 if args.nbits == 512 else EF.BAD_256)
desc = torch.zeros((40000, args.nbits // 8), dtype=torch.uint8, device="cuda")
d.computeAsync(img, kps, n=n, descriptors=desc); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters):
    d.computeAsync(img, kps, n=n, descriptors=desc)
    torch.cuda.synchronize()
print("C3 BAD%d: %d keypoints, %.4f ms per call" % (args.nbits, n, (time.perf_counter() - t0) / args.iters * 1e3))
