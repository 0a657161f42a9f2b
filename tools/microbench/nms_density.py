#!/usr/bin/env python3
"""NMS robustness cases, for rocprofv3: detect on 4K frames -- the C3/C4 frame (density 0.6, NMS radius
5), three times the default density with the default radius, the default frame -- and the 8K headline frame.
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_nms -o nms -- python tools/microbench/nms_density.py"""
import sys
import time

sys.path.insert(0, ".")
import torch

import cef_loader
from tools import synth, workloads

cef = cef_loader.load()
EF = cef.EfficientFeatures
kps = torch.zeros((5, 40000), dtype=torch.float32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
for label, rows, cols, dens, radius in (("4k_c34_radius5", 2160, 3840, 0.6, 5), ("4k_3x_density", 2160, 3840, 0.9, 15),
                                        ("4k_default", 2160, 3840, 0.3, 15), ("8k_default", 4320, 7680, 0.3, 15)):
    img = torch.from_numpy(synth.synth_frame(rows, cols, seed=1000, density=dens)).cuda()
    det = EF.create(40000, 1.2, 8, 0, 20, radius, EF.BAD_256)
    det.detectAsync(img, kps, cnt); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        det.detectAsync(img, kps, cnt)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    st = det.lastLevelStats()
    print("%-16s detect %.4f ms  corners %d  survivors %d  keypoints %d" % (label, ms, sum(x["n_candidates"] for x in st),
                                                                           sum(x["n_after_nms"] for x in st), int(cnt.item())))
