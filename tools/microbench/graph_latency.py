"""One-call latency of detectAndCompute BAD512 (8K / 4K / FHD) launched directly and replayed from a captured HIP graph
(torch.cuda.CUDAGraph around the library's launches on the capture stream).  python tools/microbench/graph_latency.py"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
for size in ("fhd", "4k", "8k"):
    rows, cols = synth.SIZES[size]
    img = torch.from_numpy(synth.synth_frame(rows, cols, seed=1000)).cuda()
    det = EF.create(40000, dtype=EF.BAD_512)
    kps = torch.zeros((5, 40000), dtype=torch.float32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    desc = torch.zeros((40000, 64), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): det.detectAndComputeAsync(img, kps, desc, cnt, stream=s)
        s.synchronize()
        ref_k, ref_d, n = kps.clone(), desc.clone(), int(cnt.item())
        t0 = time.perf_counter()
        for _ in range(50):
            det.detectAndComputeAsync(img, kps, desc, cnt, stream=s); s.synchronize()
        t_direct = (time.perf_counter() - t0) / 50 * 1e3
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            det.detectAndComputeAsync(img, kps, desc, cnt, stream=s)
        kps.zero_(); desc.zero_(); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(kps, ref_k) and torch.equal(desc, ref_d) and int(cnt.item()) == n
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay(); torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 50 * 1e3
        print(f"{size}: direct {t_direct:.4f} ms, graph replay {t_graph:.4f} ms, same result {same}, n {n}")
    except Exception as e:
        print(f"{size}: direct {t_direct:.4f} ms, graph capture failed: {type(e).__name__}: {str(e)[:200]}")
