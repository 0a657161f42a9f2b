#!/usr/bin/env python3
"""Host time of one detectAndComputeAsync call (no wait): is the enqueue the bottleneck of the three-stream bench loop?
python tools/microbench/host_enqueue_call.py [8k|4k|fhd]"""
import sys, time
sys.path.insert(0, ".")
import torch
import cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
size = sys.argv[1] if len(sys.argv) > 1 else "8k"
rows, cols = synth.SIZES[size]
nf = 40000
img = torch.from_numpy(synth.synth_frame(rows, cols, seed=1000)).cuda()
det = EF.create(nf, dtype=EF.BAD_512)
kps = torch.zeros((5, nf), dtype=torch.float32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
desc = torch.zeros((nf, 64), dtype=torch.uint8, device="cuda")
for _ in range(5):
    det.detectAndComputeAsync(img, kps, desc, cnt); torch.cuda.synchronize()
for burst in (1, 2, 4, 8):
    ts = []
    for rep in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(burst):
            det.detectAndComputeAsync(img, kps, desc, cnt)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) / burst * 1e6, (t2 - t0) / burst * 1e6))
    ts.sort()
    print("burst %d: host enqueue %.1f us per call (median), enqueue + wait %.1f us per call" % (burst, ts[len(ts) // 2][0], sorted(t[1] for t in ts)[len(ts) // 2]))
