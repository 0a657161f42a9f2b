#!/usr/bin/env python3
"""FHD / 4K detectAndCompute BAD512 under rocprofv3 (latency-bound sizes): python tools/microbench/fhd_prof.py [fhd|4k] [nfeatures]"""
import sys
sys.path.insert(0, ".")
import torch
import cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
size = sys.argv[1] if len(sys.argv) > 1 else "fhd"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
rows, cols = synth.SIZES[size]
img = torch.from_numpy(synth.synth_frame(rows, cols, seed=1000)).cuda()
det = EF.create(nf, dtype=EF.BAD_512)
kps = torch.zeros((5, nf), dtype=torch.float32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
desc = torch.zeros((nf, 64), dtype=torch.uint8, device="cuda")
for _ in range(21):
    det.detectAndComputeAsync(img, kps, desc, cnt); torch.cuda.synchronize()
print(size, nf, int(cnt.item()))
