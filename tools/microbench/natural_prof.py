#!/usr/bin/env python3
"""8K detectAndCompute BAD512 on the natural-statistics frame (1/f^beta octave noise, tools/synth.powerlaw_frames_tiled), for rocprofv3:
python tools/microbench/natural_prof.py [beta=1.3] [iters=12]"""
import sys
sys.path.insert(0, ".")
import torch
import cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
beta = float(sys.argv[1]) if len(sys.argv) > 1 else 1.3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
(f,) = synth.powerlaw_frames_tiled(4320, 7680, seed=1000, betas=(beta,))
img = torch.from_numpy(f).cuda()
det = EF.create(40000, dtype=EF.BAD_512)
for _ in range(iters):
    try:
        kps, desc, cnt = det.detectAndComputeAsync(img); torch.cuda.synchronize(); n = int(cnt.item())
    except Exception as e:          # the arena-growth frame of a dense image
        n = -1
print("beta", beta, "keypoints", n)
