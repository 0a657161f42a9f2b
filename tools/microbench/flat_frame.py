"""Kernel times on an 8K frame without corners (fixed per-tile overheads of the detector kernels); run under rocprofv3."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
import cef_loader; cef = cef_loader.load()
img = torch.full((4320, 7680), 100, dtype=torch.uint8, device='cuda')
det = cef.EfficientFeatures.create(40000, dtype=1)
kps = torch.zeros((5, 40000), dtype=torch.float32, device='cuda'); cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
desc = torch.zeros((40000, 64), dtype=torch.uint8, device='cuda')
for _ in range(6):
    det.detectAndComputeAsync(img, kps, desc, cnt)
torch.cuda.synchronize()
print('keypoints', int(cnt.item()))
