"""One-call latency of 8K detectAndCompute BAD512 (call, then wait for the stream: sample_benchmark.cpp's protocol) with the level
blur inline (EFX_BLUR_FORK=0), on the side stream (3) and decided per call (unset).  GPU box, repo root."""
import os, sys, time; sys.path.insert(0, '.')
import torch, cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
img = torch.from_numpy(synth.synth_frame(4320, 7680, seed=1000)).cuda()
def make(v):
    if v is None: os.environ.pop('EFX_BLUR_FORK', None)
    else: os.environ['EFX_BLUR_FORK'] = v
    d = EF.create(40000, dtype=EF.BAD_512)                   # the knob is read when the context is created
    out = d.detectAndComputeAsync(img); torch.cuda.synchronize()
    return d, out
ctx = {v: make(v) for v in ('0', '3', None)}
for rep in range(4):
    for v, (d, (k, desc, cnt)) in ctx.items():
        ts = []
        for i in range(30):
            t0 = time.perf_counter()
            d.detectAndComputeAsync(img, keypoints=k, descriptors=desc, count=cnt)
            torch.cuda.current_stream().synchronize()
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[5:])
        print(f'EFX_BLUR_FORK={v}: median {ts[len(ts) // 2] * 1e3:.4f} ms  min {ts[0] * 1e3:.4f}  max {ts[-1] * 1e3:.4f}')
