"""How much do frames in flight overlap?  Throughput (ms per 8K frame) of detect-only, compute-only (BAD512 on 40 000
keypoints) and detectAndCompute with 1, 2, 3 streams (one context per stream), headline workload."""
import sys, time; sys.path.insert(0, '.')
import torch, cef_loader
from tools import workloads
cef = cef_loader.load(); EF = cef.EfficientFeatures
from tools import synth
img = torch.from_numpy(synth.synth_frame(4320, 7680, seed=1000)).cuda()
N = 40000
def run(kind, ns, reps=24):
    dets = [EF.create(N, dtype=EF.BAD_512) for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    outs = []
    for d in dets:
        k, desc, cnt = d.detectAndComputeAsync(img); torch.cuda.synchronize()
        outs.append((k, desc, cnt, int(cnt.item())))
    def one(i):
        d, s = dets[i % ns], streams[i % ns]
        k, desc, cnt, n = outs[i % ns]
        if kind == 'detect': d.detectAsync(img, keypoints=k, count=cnt, stream=s)
        elif kind == 'compute': d.computeAsync(img, k, n=n, descriptors=desc, stream=s)
        else: d.detectAndComputeAsync(img, keypoints=k, descriptors=desc, count=cnt, stream=s)
    for i in range(2 * ns): one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps): one(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    run.enqueue_ms = (t1 - t0) / reps * 1e3                 # host time to enqueue one frame's launches
    return (time.perf_counter() - t0) / reps * 1e3
for kind in ('detect', 'compute', 'both'):
    print(kind, ' '.join(f'{ns} streams {run(kind, ns):.4f} ms (host enqueue {run.enqueue_ms:.4f})' for ns in (1, 2, 3)))
