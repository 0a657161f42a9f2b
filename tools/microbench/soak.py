"""Determinism soak of the headline path: three contexts / streams, the same 8 frames over and over; every result is
compared on the device with the first result of its frame (count, keypoint matrix, descriptors).  usage: soak.py <seconds>"""
import sys, time; sys.path.insert(0, '.')
import torch, cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
frames = [torch.from_numpy(synth.synth_frame(4320, 7680, seed=1000 + k)).cuda() for k in range(8)]
dets = [EF.create(40000, dtype=EF.BAD_512) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
ref = {}
for k, f in enumerate(frames):
    kps, desc, cnt = dets[0].detectAndComputeAsync(f); torch.cuda.synchronize()
    n = int(cnt.item()); ref[k] = (n, kps[:, :n].clone(), desc[:n].clone())
bad = torch.zeros(1, dtype=torch.int64, device='cuda')
outs = [None] * 3
t0 = time.time(); it = 0
while time.time() - t0 < secs:
    for j in range(24):
        i = it * 24 + j
        d, s, k = dets[i % 3], streams[i % 3], (i // 3 + i) % 8
        with torch.cuda.stream(s):
            kps, desc, cnt = d.detectAndComputeAsync(frames[k], stream=s)
            n, rk, rd = ref[k]
            bad += (cnt.to(torch.int64).sum() != n).to(torch.int64) + (kps[:, :n] != rk).any().to(torch.int64) + (desc[:n] != rd).any().to(torch.int64)
    torch.cuda.synchronize(); it += 1
print('frames', it * 24, 'mismatching results', int(bad.item()), 'in', round(time.time() - t0, 1), 's')
