#!/usr/bin/env python3
"""detectAndCompute of one synthetic frame, for profilers: python tools/microbench/dac_run.py [fhd|4k|8k] [BAD_256|BAD_512|HASH_SIFT_256|HASH_SIFT_512] [iters]"""
import sys
sys.path.insert(0, ".")
import torch
import cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
size = sys.argv[1] if len(sys.argv) > 1 else "fhd"
dt = getattr(EF, sys.argv[2] if len(sys.argv) > 2 else "BAD_512")
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rows, cols = synth.SIZES[size]
img = torch.from_numpy(synth.synth_frame(rows, cols, seed=1000)).cuda()
det = EF.create(40000, dtype=dt)
for _ in range(iters):
    kps, desc, cnt = det.detectAndComputeAsync(img); torch.cuda.synchronize()
print(size, sys.argv[2:] , int(cnt.item()))
