#!/usr/bin/env python3
"""Frames per second of detectAndCompute BAD512 with several frames in flight at FHD / 4K / 8K (one context + stream per frame in
flight, eight distinct resident frames round-robin): python tools/microbench/size_throughput.py [fhd|4k|8k] [streams] [seconds] [--graph]
--graph: every context's call is captured ONCE into a HIP graph on its stream (fixed input / output buffers); a frame is then one
device copy into the input buffer + one graph replay -- two host calls instead of a dozen launches.
Also prints the host's enqueue time per frame (the calls are asynchronous: when it approaches the GPU time per frame the host is the limit)."""
import sys, time
sys.path.insert(0, ".")
import torch
import cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
size = ([a for a in sys.argv[1:] if not a.startswith("--")] + ["fhd"])[0]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
ns = int(args[1]) if len(args) > 1 else 3
secs = float(args[2]) if len(args) > 2 else 2.0
rows, cols = synth.SIZES[size]
frames = [torch.from_numpy(synth.synth_frame(rows, cols, seed=1000 + k)).cuda() for k in range(8)]
dets = [EF.create(40000, dtype=EF.BAD_512) for _ in range(ns)]
streams = [torch.cuda.Stream() for _ in range(ns)]
outs = [(torch.zeros((5, 40000), dtype=torch.float32, device="cuda"), torch.zeros((40000, 64), dtype=torch.uint8, device="cuda"),
         torch.zeros(1, dtype=torch.int32, device="cuda")) for _ in range(ns)]
use_graph = "--graph" in sys.argv
graphs, inbuf = [], []
if use_graph:
    for j in range(ns):
        buf = frames[0].clone(); inbuf.append(buf)
        with torch.cuda.stream(streams[j]):
            dets[j].detectAndComputeAsync(buf, outs[j][0], outs[j][1], outs[j][2], stream=streams[j])      # first-use allocations
            streams[j].synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[j]):
            dets[j].detectAndComputeAsync(buf, outs[j][0], outs[j][1], outs[j][2], stream=streams[j])
        graphs.append(g)
def run(n):
    t_enq = 0.0
    for i in range(n):
        j = i % ns
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[j]):
            if use_graph:
                inbuf[j].copy_(frames[i % 8], non_blocking=True)
                graphs[j].replay()
            else:
                dets[j].detectAndComputeAsync(frames[i % 8], outs[j][0], outs[j][1], outs[j][2], stream=streams[j])
        t_enq += time.perf_counter() - t0
    torch.cuda.synchronize()
    return t_enq
run(3 * ns)
n = 240
t0 = time.perf_counter(); run(n); dt = time.perf_counter() - t0
n = max(240, int(n * secs / dt) // (8 * ns) * (8 * ns))
t0 = time.perf_counter(); enq = run(n); dt = time.perf_counter() - t0
print(f"{size} x {ns} in flight{' (graph replay)' if use_graph else ''}: {n} frames in {dt:.3f} s = {n / dt:.0f} frames/s = {dt / n * 1e3:.4f} ms/frame; host enqueue {enq / n * 1e3:.4f} ms/frame; keypoints of the last frame {int(outs[(n - 1) % ns][2].item())}")
