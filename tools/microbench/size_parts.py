"""Per-stage times (library event pairs) and the synchronous per-call latency of detectAndCompute BAD512 on FHD / 4K / 8K."""
import sys, time; sys.path.insert(0, '.')
import torch, cef_loader
from tools import synth
cef = cef_loader.load()
names = {0: 'fast', 1: 'harris', 2: 'nms', 3: 'sel+emit+ang', 10: 'describe'}
for tag, (r, c) in (('fhd', (1080, 1920)), ('4k', (2160, 3840)), ('8k', (4320, 7680))):
    det = cef.EfficientFeatures.create(40000, dtype=1)
    img = torch.from_numpy(synth.synth_frame(r, c, seed=1000)).cuda()
    kps = torch.zeros((5, 40000), dtype=torch.float32, device='cuda'); cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
    desc = torch.zeros((40000, 64), dtype=torch.uint8, device='cuda')
    for _ in range(3): det.detectAndComputeAsync(img, kps, desc, cnt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        det.detectAndComputeAsync(img, kps, desc, cnt); torch.cuda.current_stream().synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    t0 = time.perf_counter()
    for _ in range(50): det.detectAndComputeAsync(img, kps, desc, cnt)
    enq = (time.perf_counter() - t0) / 50 * 1e3
    torch.cuda.synchronize()
    det.profileEnable(64, stride=1); det.detectAndComputeAsync(img, kps, desc, cnt); torch.cuda.synchronize()
    pm, pc = det.profileRead(); parts = {}
    for m, c_ in zip(pm, pc): parts[names.get(int(c_), 'resize')] = parts.get(names.get(int(c_), 'resize'), 0) + float(m) * 1e3
    print(f'{tag}: sync call {ms:.3f} ms, host enqueue {enq:.3f} ms, keypoints {int(cnt.item())}, us: ' + ' '.join(f'{k} {v:.0f}' for k, v in parts.items()) + f' sum {sum(parts.values()):.0f}')
