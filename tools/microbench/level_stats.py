"""Prints the per-level detector statistics of one synthetic 8K frame (investigation helper)."""
import sys
sys.path.insert(0, '.')
import torch
import cef_loader; cef = cef_loader.load()
from tools import synth
img = torch.from_numpy(synth.synth_frame(4320, 7680, seed=1000)).cuda()
det = cef.EfficientFeatures.create(40000, dtype=1)
kps = torch.zeros((5, 40000), dtype=torch.float32, device='cuda'); cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
det.detectAsync(img, kps, cnt); torch.cuda.synchronize()
tot = 0
for l, s in enumerate(det.lastLevelStats()):
    r, c, _ = det.levelGeometry(4320, 7680, l)
    tiles = ((r + 63) // 64) * ((c + 63) // 64)
    tot += s['n_candidates']
    print(l, r, c, tiles, s, 'corners/tile %.1f' % (s['n_candidates'] / tiles))
print('total corners', tot)
