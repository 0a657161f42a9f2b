#!/usr/bin/env python3
"""End-to-end ms/frame from HOST frames (PCIe inclusive): upload (+ BGR->gray) + detectAndCompute BAD512 on 8K,
through the double-buffered uploader.  This is the number noted in DESIGN.md next to the
HBM-resident headline metric; it is never bench.py's `value`."""
import argparse
import ctypes as C
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import cef_loader
from tools import synth

cef = cef_loader.load()


def run(frames_host, iters, nstream=2):
    rows, cols = frames_host[0].shape[:2]
    ups = [cef.Uploader() for _ in range(nstream)]
    dets = [cef.EfficientFeatures.create(40000, dtype=cef.EfficientFeatures.BAD_512) for _ in range(nstream)]
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    kps = [torch.zeros((5, 40000), dtype=torch.float32, device="cuda") for _ in range(nstream)]
    desc = [torch.zeros((40000, 64), dtype=torch.uint8, device="cuda") for _ in range(nstream)]
    cnt = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(nstream)]

    def one(i):
        s = i % nstream
        d, pitch, r, c = ups[s].upload(frames_host[i % len(frames_host)], stream=streams[s])
        rc = cef.lib().efx_detect_and_compute_async(dets[s]._h, C.c_void_p(d), r, c, C.c_size_t(pitch), C.c_void_p(kps[s].data_ptr()),
                                                    C.c_size_t(kps[s].stride(0) * 4), C.c_void_p(desc[s].data_ptr()), C.c_size_t(64),
                                                    40000, C.c_void_p(cnt[s].data_ptr()), C.c_void_p(streams[s].cuda_stream))
        assert rc == 0
    for i in range(4):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        one(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, int(cnt[0].item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=32)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rows, cols = synth.SIZES["8k"]
    gray = [synth.synth_frame(rows, cols, seed=1000 + k) for k in range(2)]
    res = {"workload": "8K host frame -> upload -> detectAndCompute BAD512 (40k kp), 2 streams, double-buffered uploader",
           "device": torch.cuda.get_device_name(0), "rows": []}
    for label, ch in (("gray", 1), ("BGR", 3), ("BGRA", 4)):
        if ch == 1:
            src = gray
        else:
            src = [np.ascontiguousarray(np.repeat(g[:, :, None], ch, axis=2)) for g in gray]
        pinned = []
        for f in src:
            h = cef.host_alloc(f.shape); h[...] = f; pinned.append(h)
        for mem, frames in (("pinned", pinned), ("pageable", src)):
            ms, n = run(frames, a.iters)
            mb = src[0].nbytes / 1e6
            res["rows"].append({"input": label, "host_memory": mem, "ms_per_frame": round(ms, 3), "keypoints": n,
                                "Mkeypoints_per_s": round(n / ms / 1e3, 2), "upload_MB": round(mb, 1),
                                "effective_upload_GBps_if_bound": round(mb / ms, 1)})
        for h in pinned:
            cef.host_free(h)
    s = json.dumps(res, indent=1)
    print(s)
    if a.out:
        open(a.out, "w").write(s)


if __name__ == "__main__":
    main()
