"""detectAndCompute BAD512 time on 4K frames of different corner densities, incl. pure noise (10 % cap active)."""
import sys, time; sys.path.insert(0, '.')
import torch, cef_loader
from tools import synth
cef = cef_loader.load()
det = cef.EfficientFeatures.create(40000, dtype=1)
kps = torch.zeros((5, 40000), dtype=torch.float32, device='cuda'); cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
desc = torch.zeros((40000, 64), dtype=torch.uint8, device='cuda')
cases = [('density 0.05', synth.synth_frame(2160, 3840, seed=1, density=0.05)), ('density 0.3', synth.synth_frame(2160, 3840, seed=1)),
         ('density 1.0', synth.synth_frame(2160, 3840, seed=1, density=1.0)), ('density 3.0', synth.synth_frame(2160, 3840, seed=1, density=3.0)),
         ('noise (cap)', synth.noise_frame(2160, 3840, seed=1)), ('flat', synth.noise_frame(2160, 3840, seed=1) * 0 + 77)]
for name, im in cases:
    img = torch.from_numpy(im).cuda()
    for _ in range(2): det.detectAndComputeAsync(img, kps, desc, cnt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): det.detectAndComputeAsync(img, kps, desc, cnt)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    st = det.lastLevelStats()
    det.profileEnable(64, stride=1); det.detectAndComputeAsync(img, kps, desc, cnt); torch.cuda.synchronize()
    pm, pc = det.profileRead(); names = {0: 'fast', 1: 'harris', 2: 'nms', 3: 'sel+emit+ang', 10: 'describe'}
    parts = {}
    for m, c in zip(pm, pc): parts[names.get(int(c), 'resize')] = parts.get(names.get(int(c), 'resize'), 0) + float(m) * 1e3
    print(f'{name:14s} {ms:7.3f} ms  keypoints {int(cnt.item()):6d}  FAST corners {sum(s["n_candidates"] for s in st):8d}  survivors {sum(s["n_after_nms"] for s in st):7d}  us: ' + ' '.join(f'{k} {v:.0f}' for k, v in parts.items()))
