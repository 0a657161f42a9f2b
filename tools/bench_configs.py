#!/usr/bin/env python3
"""Measures the BASELINE.json configurations next to the headline metric (which bench.py owns):
  C2  4K detect-only                                   (README.md:52-54 protocol)
  C3  4K, 40k keypoints, compute-only BAD256 / BAD512  (README.md:60-62; sample_benchmark.cpp:132-141)
  C4  4K, 40k keypoints, compute-only HashSIFT256/512
      (tools/workloads.py: the C3 / C4 keypoints are the detector's on a denser 4K frame with NMS radius 5 -- EXACTLY
      40 000 over all eight levels; the default radius saturates a 4K pyramid at ~22 000)
  plus the rows of the reference's README tables: detect on FHD / 4K / 8K, compute and detectAndCompute on 8K for all
  four descriptor types.
Protocol of samples/sample_benchmark.cpp:39-52: 1 warm-up + N timed iterations of the async call followed by a
stream synchronise, input resident on the device; every iteration is timed on its own, so a row carries the protocol's
mean AND min / median.

`measure()` is what `bench.py --gpus 1` puts into its JSON line as `configs` (the driver-run line then
holds C2 / C3 / C4 and the README rows, each with its roofline -- SURVEY 8d's bytes AND this design's own bytes -- and, for
C2 / C3 / C4, the oracle on one host thread beside it); run as a script it prints the same object (--out writes it)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tools import synth, workloads

HBM_GBS = 8000.0
# README.md:52-71 of the reference, RTX 3060 Ti, milliseconds: OTHER hardware, quoted for orientation
README_MS = {("detect", "fhd"): 1.6, ("detect", "4k"): 2.9, ("detect", "8k"): 5.5,
             ("compute", "BAD256"): 1.5, ("compute", "BAD512"): 2.7, ("compute", "HashSIFT256"): 3.5, ("compute", "HashSIFT512"): 3.9,
             ("detectAndCompute", "BAD256"): 7.2, ("detectAndCompute", "BAD512"): 8.2, ("detectAndCompute", "HashSIFT256"): 8.5,
             ("detectAndCompute", "HashSIFT512"): 8.9}


def perf(torch, fn, iters):
    """1 warm-up + `iters` x (call + device synchronise), each iteration timed: mean (the protocol's figure), min, median."""
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) * 1e3)
    t = np.array(t)
    return {"ms": round(float(t.mean()), 4), "ms_min": round(float(t.min()), 4), "ms_median": round(float(np.median(t)), 4), "iters": iters}


def cpu_time(fn, repeats=3):
    t = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        r = fn()
        t.append(time.perf_counter() - t0)
    return r, {"s_min": round(min(t), 4), "s_median": round(float(np.median(t)), 4), "repeats": repeats}


def roof(bytes_survey, bytes_design, ms, note=None):
    r = {"bound": "hbm", "peak_GBps": HBM_GBS,
         "survey_8d_MB": round(bytes_survey / 1e6, 1), "frac_vs_survey_8d_bytes": round(bytes_survey / ms / 1e6 / HBM_GBS, 4),
         "design_MB": round(bytes_design / 1e6, 1), "frac_vs_design_bytes": round(bytes_design / ms / 1e6 / HBM_GBS, 4)}
    if note:
        r["note"] = note
    return r


def detect_bytes(det, rows, cols, stats):
    """(2F - 1) P for detect.  This design: the resize chain reads / writes every level once, FAST and Harris
    each read the levels once more, corner and survivor records go through HBM."""
    P = [det.levelGeometry(rows, cols, l)[0] * det.levelGeometry(rows, cols, l)[1] for l in range(8)]
    C = sum(s["n_candidates"] for s in stats)
    S = sum(s["n_after_nms"] for s in stats)
    survey = sum(P) + sum(P[1:])
    design = (sum(P[:-1]) + sum(P[1:])) + (sum(P) + 4 * C) + (sum(P) + 8 * C) + (8 * C + 8 * S)
    return survey, design, sum(P)


def measure(cef, iters=30, cpu_baseline=True, sizes=("fhd", "4k", "8k"), log=lambda s: None, data_rows=True, batch_rows=True):
    import torch
    EF = cef.EfficientFeatures
    types = (("BAD256", EF.BAD_256, 32), ("BAD512", EF.BAD_512, 64), ("HashSIFT256", EF.HASH_SIFT_256, 32), ("HashSIFT512", EF.HASH_SIFT_512, 64))
    res = {"protocol": "sample_benchmark.cpp:39-52: 1 warm-up + %d x (async call + device synchronise), input on the device, one stream, "
                       "nfeatures = 40000; ms = mean (the protocol's figure), ms_min / ms_median beside it" % iters,
           "device": torch.cuda.get_device_name(0),
           "readme_ms": "the reference's README on an RTX 3060 Ti (other hardware, BASELINE.md section 1)", "rows": []}
    oracle = None
    if cpu_baseline:
        from oracle import pyoracle as oracle
        oracle.set_threads(1)
        res["cpu_baseline"] = {"kind": "port", "cores": 1, "what": "oracle/efx_oracle.c on one host thread (the reference CPU module is "
                               "single-threaded as written), 3 repeats, min / median in seconds", "host_cores_available": os.cpu_count()}
    kps = torch.zeros((5, workloads.N40K), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")

    # ---- C2: 4K detect-only
    log("C2")
    f2 = workloads.frame_c2()
    img = torch.from_numpy(f2).cuda()
    det = EF.create(workloads.N40K, dtype=EF.BAD_256)
    t = perf(torch, lambda: det.detectAsync(img, kps, cnt), iters)
    n = int(cnt.item())
    bs, bd, _ = detect_bytes(det, workloads.K4[0], workloads.K4[1], det.lastLevelStats())
    row = dict(config="C2", what="4K frame, detect-only (pyramid + FAST-9 + Harris + radius NMS + quota + IC angle)", keypoints=n, **t,
               Mkeypoints_per_s=round(n / t["ms"] / 1e3, 2), readme_ms=README_MS[("detect", "4k")], roofline=roof(bs, bd, t["ms"]))
    if oracle:
        ref, ct = cpu_time(lambda: oracle.detect_and_compute(f2, nfeatures=workloads.N40K, desc_type=-1))
        row["cpu_baseline"] = dict(ct, keypoints=ref["n"], Mkeypoints_per_s=round(ref["n"] / ct["s_min"] / 1e6, 5), sample="the whole frame")
    res["rows"].append(row)
    del det

    # ---- C3 / C4: compute-only on EXACTLY 40 000 keypoints of a 4K frame
    f34 = workloads.frame_c34()
    img = torch.from_numpy(f34).cuda()
    rows, cols = workloads.K4
    det = EF.create(workloads.N40K, 1.2, 8, 0, 20, workloads.C34_NMS_RADIUS, EF.BAD_256)
    det.detectAsync(img, kps, cnt); torch.cuda.synchronize()
    n = int(cnt.item())
    kp4 = None
    if oracle:
        k = cef.unpack_keypoints(kps[:, :n].cpu().numpy())
        kp4 = np.stack([k["x"].astype(np.float32), k["y"].astype(np.float32), np.full(n, 31, np.float32), k["angle"]], axis=1)
    del det
    px, integ = rows * cols, 4 * (rows + 1) * (cols + 1)
    for name, dt, nbytes in types:
        cfg = "C3" if name.startswith("BAD") else "C4"
        log(cfg + " " + name)
        d = EF.create(workloads.N40K, dtype=dt)
        desc = torch.zeros((workloads.N40K, nbytes), dtype=torch.uint8, device="cuda")
        t = perf(torch, lambda: d.computeAsync(img, kps, n=n, descriptors=desc), iters)
        row = dict(config=cfg, what="4K frame, compute-only %s on the detector's keypoints (NMS radius %d, frame density %.1f: tools/workloads.py)"
                   % (name, workloads.C34_NMS_RADIUS, workloads.C34_DENSITY), descriptor=name, keypoints=n, **t,
                   Mdescriptors_per_s=round(n / t["ms"] / 1e3, 2), readme_ms=README_MS[("compute", name)])
        if cfg == "C3":
            # SURVEY 8d BAD compute (the reference's design): P + 4 (W+1)(H+1) written + min(that, N 46^2 4) gathered + 16 N + N nbits/8;
            # this design: the 48 x 48 u8 windows lie inside the frame (P), 20 B of keypoint matrix in, the descriptor out
            row["roofline"] = roof(px + integ + min(integ, n * 46 * 46 * 4) + 16 * n + n * nbytes, px + 20 * n + n * nbytes, t["ms"],
                                   "bad_raw_kernel is bound by LDS-array cycles (random box gathers) and VALU issue, not by HBM (DESIGN.md section 5)")
            cpu = (lambda nb=nbytes * 8: oracle.bad_compute(f34, kp4, nb)) if oracle else None
            ncpu = n
        else:
            flop = 2.0 * 129 * nbytes * 8 * n
            row["roofline"] = {"bound": "mfma (projection only)", "projection_GFLOP": round(flop / 1e9, 3), "peak_TFLOPs_bf16_dense": 2500.0,
                               "note": "the call is dominated by patch_sift_kernel (VALU issue + LDS atomics), the projection runs as a 3-term "
                                       "exact bf16 split on v_mfma_f32_32x32x16_bf16 (DESIGN.md section 5)",
                               "design_MB": round((px + 20 * n + n * (288 + 64) * 2 + n * nbytes) / 1e6, 1),
                               "frac_vs_design_bytes": round((px + 20 * n + n * (288 + 64) * 2 + n * nbytes) / t["ms"] / 1e6 / HBM_GBS, 4)}
            ncpu = 10000                                      # bounded sample: the CPU HashSIFT takes ~0.1 ms per keypoint
            cpu = (lambda nb=nbytes * 8: oracle.hashsift_compute(f34, kp4[:ncpu], nb)) if oracle else None
        if cpu:
            _, ct = cpu_time(cpu)
            row["cpu_baseline"] = dict(ct, keypoints=ncpu, Mdescriptors_per_s=round(ncpu / ct["s_min"] / 1e6, 5),
                                       sample="all %d keypoints" % ncpu if ncpu == n else "the first %d of the %d keypoints" % (ncpu, n))
        res["rows"].append(row)
        del d

    # ---- the README's rows: detect FHD / 4K / 8K; compute and detectAndCompute at 8K (BASELINE.md's reading of the tables)
    for size in sizes:
        r_, c_ = synth.SIZES[size]
        img = torch.from_numpy(synth.synth_frame(r_, c_, seed=1000)).cuda()
        det = EF.create(workloads.N40K, dtype=EF.BAD_256)
        log("detect " + size)
        t = perf(torch, lambda: det.detectAsync(img, kps, cnt), iters)
        n = int(cnt.item())
        bs, bd, sumP = detect_bytes(det, r_, c_, det.lastLevelStats())
        res["rows"].append(dict(config="readme", mode="detect", size=size, keypoints=n, **t, Mkeypoints_per_s=round(n / t["ms"] / 1e3, 2),
                                readme_ms=README_MS[("detect", size)], roofline=roof(bs, bd, t["ms"])))
        del det
        for name, dt, nbytes in types:
            if size != "8k" and name not in ("BAD512", "HashSIFT512"):
                continue
            log("%s %s" % (size, name))
            d = EF.create(workloads.N40K, dtype=dt)
            desc = torch.zeros((workloads.N40K, nbytes), dtype=torch.uint8, device="cuda")
            d.detectAsync(img, kps, cnt); torch.cuda.synchronize()
            n = int(cnt.item())
            tc = perf(torch, lambda: d.computeAsync(img, kps, n=n, descriptors=desc), iters)
            tdc = perf(torch, lambda: d.detectAndComputeAsync(img, kps, desc, cnt), iters)
            row = dict(config="readme", mode="compute + detectAndCompute", size=size, descriptor=name, keypoints=n,
                       compute=dict(tc, Mdescriptors_per_s=round(n / tc["ms"] / 1e3, 2)),
                       detectAndCompute=dict(tdc, Mkeypoints_per_s=round(n / tdc["ms"] / 1e3, 2)))
            if size == "8k":
                row["compute"]["readme_ms"] = README_MS[("compute", name)]
                row["detectAndCompute"]["readme_ms"] = README_MS[("detectAndCompute", name)]
            if name.startswith("BAD"):
                # detectAndCompute, SURVEY 8d: detect + per-level blur (2 F P) + global int32 integrals (5 F P) + gathers ~ 0.9 GB at 8K;
                # this design: detect's own bytes + the windows inside the levels (F P) + 80 B record in + descriptor out per keypoint
                st = d.lastLevelStats()
                bs2, bd2, _ = detect_bytes(d, r_, c_, st)
                row["detectAndCompute"]["roofline"] = roof(bs2 + 7 * sumP + 46 * 46 * 4 * n + n * nbytes, bd2 + sumP + (80 + nbytes) * n, tdc["ms"])
            res["rows"].append(row)
            del d
    # ---- data dependence of the headline: the same call on frames with the statistics of photographs.  The
    # headline's rectangle frames have 2.8 M FAST corners per 8K frame (1 pixel in 37); a 1/f^1.3 texture has ~3 % corners, a 1/f^1.0
    # one ~17 % -- the reference's 10 % candidate cap (.cpp:252, spec S2) is active.  Since round 6 the FIRST call on such a frame is
    # complete (first_call_keypoints; rounds 2-5: void once, then the context regrew to 2.1 GB)
    if "8k" in sizes and data_rows:
        r_, c_ = synth.SIZES["8k"]
        log("natural-statistics frames")
        betas = (1.3, 1.0)
        for beta, f in zip(betas, synth.powerlaw_frames_tiled(r_, c_, seed=1000, betas=betas)):
            img = torch.from_numpy(f).cuda()
            d = EF.create(workloads.N40K, dtype=EF.BAD_512)
            desc = torch.zeros((workloads.N40K, 64), dtype=torch.uint8, device="cuda")
            d.detectAndComputeAsync(img, kps, desc, cnt); torch.cuda.synchronize()
            first_n = int(cnt.item())                         # the context's very first call
            t = perf(torch, lambda: d.detectAndComputeAsync(img, kps, desc, cnt), iters)
            n = int(cnt.item())
            st = d.lastLevelStats()
            C = sum(x["n_candidates"] for x in st); S = sum(x["n_after_nms"] for x in st)
            bs2, bd2, sumP = detect_bytes(d, r_, c_, st)
            res["rows"].append(dict(config="data", mode="detectAndCompute", size="8k", descriptor="BAD512",
                                    frame="1/f^%.1f octave noise (tools/synth.py powerlaw_frames_tiled, seed 1000)" % beta,
                                    fast_corners=int(C), fast_corners_frac_of_pixels=round(C / sumP, 4), nms_survivors=int(S), keypoints=n,
                                    cap_active=bool(any(x["n_candidates"] >= 0.1 * x2 for x, x2 in zip(st, [d.levelGeometry(r_, c_, l)[0] * d.levelGeometry(r_, c_, l)[1] for l in range(8)]))),
                                    first_call_keypoints=first_n, overflow_events=int(d.overflowEvents()) if hasattr(d, "overflowEvents") else None,
                                    device_MB=round(d.deviceBytes() / 1e6, 1) if hasattr(d, "deviceBytes") else None,
                                    **t, Mkeypoints_per_s=round(n / t["ms"] / 1e3, 2),
                                    roofline=roof(bs2 + 7 * sumP + 46 * 46 * 4 * n + n * 64, bd2 + sumP + (80 + 64) * n, t["ms"])))
            del d
    # ---- frame-batched launches (round 6; SURVEY 8b, samples/sample_image_sequence.cpp:70-105): B same-sized frames per launch
    # chain on each of two contexts / streams, distinct resident frames, against the same entry point as a loop of single-frame
    # calls (EFX_NO_BATCH=1, four contexts: the best per-frame form of round 5)
    if batch_rows:
        def frames_per_s(size, B, nctx, seconds=1.0):
            r_, c_ = synth.SIZES[size]
            nd = min(B * nctx, 16)
            base = [torch.from_numpy(synth.synth_frame(r_, c_, seed=1000 + k)).cuda() for k in range(nd)]
            Fn = B * nctx
            fr = [base[i % nd] for i in range(Fn)]
            dets = [EF.create(workloads.N40K, dtype=EF.BAD_512) for _ in range(nctx)]
            sts = [torch.cuda.Stream() for _ in range(nctx)]
            kk = [torch.zeros((5, workloads.N40K), dtype=torch.float32, device="cuda") for _ in range(Fn)]
            dd = [torch.zeros((workloads.N40K, 64), dtype=torch.uint8, device="cuda") for _ in range(Fn)]
            cc = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(Fn)]
            b = cef.Batch(dets, sts, fr, kk, dd, cc, workloads.N40K)
            for _ in range(3):
                b.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter(); nrun = 0
            while True:
                for _ in range(10):
                    b.run()
                nrun += 10
                torch.cuda.synchronize()
                if time.perf_counter() - t0 >= seconds:
                    break
            dt = time.perf_counter() - t0
            nk = int(sum(int(c.item()) for c in cc))
            out = dict(frames_per_s=round(nrun * Fn / dt, 1), ms_per_frame=round(dt / (nrun * Fn) * 1e3, 5), keypoints_per_frame=round(nk / Fn, 1),
                       Mpx_per_s=round(nrun * Fn * r_ * c_ / dt / 1e6, 1), device_MB_per_context=round(dets[0].deviceBytes() / 1e6, 1))
            del b, dets
            return out
        for size, B in (("fhd", 16), ("4k", 8), ("8k", 2)):
            if size not in sizes:
                continue
            log("batch " + size)
            row = dict(config="batch", mode="detectAndCompute", size=size, descriptor="BAD512", frames_per_launch=B, contexts=2,
                       batched=frames_per_s(size, B, 2))
            os.environ["EFX_NO_BATCH"] = "1"
            try:
                row["per_frame_calls"] = dict(frames_per_s(size, 1, 4), contexts=4)
            finally:
                del os.environ["EFX_NO_BATCH"]
            res["rows"].append(row)
    if oracle:
        oracle.set_threads(1)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--out", default="")
    ap.add_argument("--sizes", default="fhd,4k,8k")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    import cef_loader
    cef = cef_loader.load()
    res = measure(cef, args.iters, not args.no_cpu_baseline, tuple(args.sizes.split(",")), log=lambda s: print("..", s, file=sys.stderr, flush=True))
    s = json.dumps(res, indent=1)
    print(s)
    if args.out:
        open(args.out, "w").write(s)


if __name__ == "__main__":
    main()
