#!/usr/bin/env python3
"""Frames per second of detectAndCompute BAD512 through FRAME-BATCHED launches (round 6): B frames per launch chain on each of
`nctx` contexts / streams (efx_detect_and_compute_batch_async; B * nctx distinct resident frames), against the per-frame form
(EFX_NO_BATCH=1 in the environment: the same entry point as a loop of single-frame calls).
python tools/microbench/batch_throughput.py [fhd|4k|8k] [B] [nctx] [seconds] [nfeatures] [BAD_512|BAD_256|HASH_SIFT_512|HASH_SIFT_256]"""
import sys, time
sys.path.insert(0, ".")
import torch
import cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
a = sys.argv[1:]
size = a[0] if len(a) > 0 else "fhd"
B = int(a[1]) if len(a) > 1 else 16
nctx = int(a[2]) if len(a) > 2 else 2
secs = float(a[3]) if len(a) > 3 else 2.0
nf = int(a[4]) if len(a) > 4 else 40000
dname = a[5] if len(a) > 5 else "BAD_512"
dtype = getattr(EF, dname); nbytes = 64 if dname.endswith("512") else 32
rows, cols = synth.SIZES[size]
ndist = min(B * nctx, 16)
base = [torch.from_numpy(synth.synth_frame(rows, cols, seed=1000 + k)).cuda() for k in range(ndist)]
F = B * nctx
frames = [base[i % ndist] for i in range(F)]
dets = [EF.create(nf, dtype=dtype) for _ in range(nctx)]
streams = [torch.cuda.Stream() for _ in range(nctx)]
kps = [torch.zeros((5, nf), dtype=torch.float32, device="cuda") for _ in range(F)]
desc = [torch.zeros((nf, nbytes), dtype=torch.uint8, device="cuda") for _ in range(F)]
cnt = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(F)]
batch = cef.Batch(dets, streams, frames, kps, desc, cnt, nf)
def run(n):
    t = 0.0
    for _ in range(n):
        t0 = time.perf_counter(); batch.run(); t += time.perf_counter() - t0
    torch.cuda.synchronize()
    return t
run(3)
n = 20
t0 = time.perf_counter(); run(n); dt = time.perf_counter() - t0
n = max(20, int(n * secs / dt))
t0 = time.perf_counter(); enq = run(n); dt = time.perf_counter() - t0
nfr = n * F
print(f"{size} {dname} B={B} x {nctx} contexts: {nfr} frames in {dt:.3f} s = {nfr / dt:.0f} frames/s = {dt / nfr * 1e3:.4f} ms/frame; host enqueue {enq / nfr * 1e3:.4f} ms/frame; "
      f"keypoints of the last frame {int(cnt[-1].item())}; context {dets[0].deviceBytes() / 1e6:.0f} MB")
