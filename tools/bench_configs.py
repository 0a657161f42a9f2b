#!/usr/bin/env python3
"""Measures the BASELINE.json configurations next to the headline metric (which bench.py owns):
  C2  4K detect-only                                   (README.md:52-54 protocol)
  C3  4K, 40k keypoints, compute-only BAD256 / BAD512  (README.md:60-62; sample_benchmark.cpp:132-141)
  C4  4K, 40k keypoints, compute-only HashSIFT256/512
      (tools/workloads.py: the C3 / C4 keypoints are the detector's on a denser 4K frame with NMS radius 5 -- EXACTLY
      40 000 over all eight levels; the default radius saturates a 4K pyramid at ~22 000)
  plus detect / detectAndCompute on FHD, 4K and 8K for all four descriptor types.
Protocol of samples/sample_benchmark.cpp:39-52: 1 warm-up + N timed iterations of the async call followed by a
stream synchronise, input resident on the device.  Prints one JSON object; --out writes it to a file."""
import argparse
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import cef_loader
from tools import synth, workloads

cef = cef_loader.load()
EF = cef.EfficientFeatures


def perf(fn, iters):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--out", default="")
    ap.add_argument("--sizes", default="fhd,4k,8k")
    args = ap.parse_args()
    res = {"protocol": "1 warm-up + %d iterations, async call + stream sync, input on device, nfeatures=40000" % args.iters,
           "device": torch.cuda.get_device_name(0), "rows": []}
    for size in args.sizes.split(","):
        rows, cols = synth.SIZES[size]
        img = torch.from_numpy(synth.synth_frame(rows, cols, seed=1000)).cuda()
        px = rows * cols
        kps = torch.zeros((5, 40000), dtype=torch.float32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        det = EF.create(40000, dtype=EF.BAD_256)
        ms = perf(lambda: det.detectAsync(img, kps, cnt), args.iters)
        n = int(cnt.item())
        # algorithmic bytes of detect: (2F-1) P
        lv = [det.levelGeometry(rows, cols, l) for l in range(8)]
        P = [r * c for r, c, _ in lv]
        bytes_detect = sum(P) + sum(P[1:])
        res["rows"].append({"size": size, "mode": "detect", "ms": round(ms, 4), "keypoints": n,
                            "Mkeypoints_per_s": round(n / ms / 1e3, 2),
                            "algorithmic_MB": round(bytes_detect / 1e6, 1),
                            "achieved_GBps": round(bytes_detect / ms / 1e6, 1), "frac_of_8TBps": round(bytes_detect / ms / 1e6 / 8000, 4)})
        for name, dt, nbytes in (("BAD256", EF.BAD_256, 32), ("BAD512", EF.BAD_512, 64),
                                 ("HashSIFT256", EF.HASH_SIFT_256, 32), ("HashSIFT512", EF.HASH_SIFT_512, 64)):
            d = EF.create(40000, dtype=dt)
            desc = torch.zeros((40000, nbytes), dtype=torch.uint8, device="cuda")
            d.detectAsync(img, kps, cnt); torch.cuda.synchronize()
            n = int(cnt.item())
            ms_c = perf(lambda: d.computeAsync(img, kps, n=n, descriptors=desc), args.iters)
            ms_dc = perf(lambda: d.detectAndComputeAsync(img, kps, desc, cnt), args.iters)
            row = {"size": size, "descriptor": name, "keypoints": n, "compute_ms": round(ms_c, 4),
                   "Mdescriptors_per_s": round(n / ms_c / 1e3, 2), "detectAndCompute_ms": round(ms_dc, 4),
                   "detectAndCompute_Mkeypoints_per_s": round(n / ms_dc / 1e3, 2)}
            if name.startswith("BAD"):
                # SURVEY 8d BAD compute: P + 2*4(W+1)(H+1) + 16N + N*nbits/8 (the reference's global-integral design)
                b = px + 8 * (rows + 1) * (cols + 1) + 16 * n + n * nbytes
                row["survey_algorithmic_MB"] = round(b / 1e6, 1)
                row["compute_frac_of_8TBps_vs_survey_bytes"] = round(b / ms_c / 1e6 / 8000, 4)
            else:
                flop = 2.0 * 129 * (nbytes * 8) * n
                row["projection_GFLOP"] = round(flop / 1e9, 3)
            res["rows"].append(row)
    # C3 / C4 as BASELINE.json states them: compute-only on EXACTLY 40 000 keypoints of a 4K frame
    img = torch.from_numpy(workloads.frame_c34()).cuda()
    rows, cols = workloads.K4
    kps = torch.zeros((5, 40000), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    det = EF.create(workloads.N40K, 1.2, 8, 0, 20, workloads.C34_NMS_RADIUS, EF.BAD_256)
    det.detectAsync(img, kps, cnt); torch.cuda.synchronize()
    n = int(cnt.item())
    for name, dt, nbytes, cfg in (("BAD256", EF.BAD_256, 32, "C3"), ("BAD512", EF.BAD_512, 64, "C3"),
                                  ("HashSIFT256", EF.HASH_SIFT_256, 32, "C4"), ("HashSIFT512", EF.HASH_SIFT_512, 64, "C4")):
        d = EF.create(40000, dtype=dt)
        desc = torch.zeros((40000, nbytes), dtype=torch.uint8, device="cuda")
        ms_c = perf(lambda: d.computeAsync(img, kps, n=n, descriptors=desc), args.iters)
        row = {"config": cfg, "size": "4k", "descriptor": name, "keypoints": n, "compute_ms": round(ms_c, 4),
               "Mdescriptors_per_s": round(n / ms_c / 1e3, 2),
               "keypoints_from": "detector, NMS radius %d, frame density %.1f (tools/workloads.py)" % (workloads.C34_NMS_RADIUS, workloads.C34_DENSITY)}
        if cfg == "C3":
            # roofline of the whole call (host clock, one stream, sync per call) against SURVEY 8d's BAD-compute figure --
            # P + 4 (W+1)(H+1) + min(4 (W+1)(H+1), N 46^2 4) + 16 N + N nbits/8: 77.9 MB for BAD512 -- and against what THIS
            # design has to move: the 48 x 48 u8 windows lie inside the frame (P), 20 B of keypoint matrix, the descriptor
            px = rows * cols
            integ = 4 * (rows + 1) * (cols + 1)
            b_survey = px + integ + min(integ, n * 46 * 46 * 4) + 16 * n + n * nbytes
            b_design = px + 20 * n + n * nbytes
            row["roofline"] = {"bound": "hbm", "peak_GBps": 8000,
                               "survey_algorithmic_MB": round(b_survey / 1e6, 1), "frac_vs_survey_bytes": round(b_survey / ms_c / 1e6 / 8000, 4),
                               "design_algorithmic_MB": round(b_design / 1e6, 1), "frac_vs_design_bytes": round(b_design / ms_c / 1e6 / 8000, 4),
                               "note": "the kernel is bound by LDS-array cycles (random box gathers) and VALU issue, not by HBM: profiles/r03_c3_*"}
        res["rows"].append(row)
    # NMS robustness: detect on frames whose corner density or NMS radius make the exact scans long
    for label, dens, radius in (("c34_frame_radius5", workloads.C34_DENSITY, workloads.C34_NMS_RADIUS), ("3x_density_default_radius", 0.9, 15),
                                ("default", None, 15)):
        f = workloads.frame_c2() if dens is None else synth.synth_frame(rows, cols, seed=1000, density=dens)
        d_f = torch.from_numpy(f).cuda()
        det = EF.create(workloads.N40K, 1.2, 8, 0, 20, radius, EF.BAD_256)
        ms = perf(lambda: det.detectAsync(d_f, kps, cnt), args.iters)
        st = det.lastLevelStats()
        res["rows"].append({"config": "nms_density", "case": label, "size": "4k", "nonmax_radius": radius, "detect_ms": round(ms, 4),
                            "fast_corners": int(sum(x["n_candidates"] for x in st)), "nms_survivors": int(sum(x["n_after_nms"] for x in st)),
                            "keypoints": int(cnt.item())})
    s = json.dumps(res, indent=1)
    print(s)
    if args.out:
        open(args.out, "w").write(s)


if __name__ == "__main__":
    main()
