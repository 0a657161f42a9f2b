#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X.

Metric (BASELINE.json): Mkeypoints/s of detectAndCompute on 8K frames, 40 000 keypoints requested, BAD512,
reference defaults otherwise (8 levels, scale 1.2, FAST threshold 20, NMS radius 15) -- the protocol of
samples/sample_benchmark.cpp:104-142 (input already resident on the device, async call + stream sync).

A "step" is one pass of the hot path over one batch of FRAMES_PER_STEP independent synthetic 8K frames
(seeds 1000+k, BASELINE.json configs[4]: 64 frames over 8 GPUs = 8 per GPU).  Frames shard across ranks with
no data-path collective (weak scaling); RCCL is used only to reduce the timing / keypoint counters.

Output: ONE JSON line on rank 0 (driver contract) with `roofline` (the kernel with the longest average launch among
fast / harris / nms / bad, against the HBM roofline, timed live with HIP events on the launch stream) and `cpu_baseline` (the oracle timed on the
host cores, rank 0, N=1 only, one 8K frame).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS, COLS = 4320, 7680            # 8K
NFEATURES = 40000
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# the reference's published detectAndCompute BAD512 time for this workload, 8.2 ms per frame on an
# RTX 3060 Ti (README.md:68-70) = 40000 / 8.2 ms = 4.878 Mkeypoints/s (one GPU of other hardware; the only published number)
BASELINE_MS = 8.2                                  # detectAndCompute BAD512, 8K, 40 000 kp, RTX 3060 Ti
BASELINE_MKPS = 40000 / (BASELINE_MS * 1e-3) / 1e6


def detect_algorithmic_bytes(det, rows, cols, nlevels=8):
    """pixels per level, and (every level read once + every derived level written once) per frame."""
    px = [det.levelGeometry(rows, cols, l)[0] * det.levelGeometry(rows, cols, l)[1] for l in range(nlevels)]
    return px, float(sum(px) + sum(px[1:]))


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run on this
    node, exactly as the driver does, and pass their exit status on.  Fails loudly when the node has fewer GPUs."""
    import subprocess
    if not args.dry_run:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} requested but {have} GPU(s) visible on this node", file=sys.stderr, flush=True)
            raise SystemExit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this host driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    raise SystemExit(subprocess.call(cmd, env=env))


def collective_world(dist, one):
    """Number of ranks that took part in a real all-reduce (SUM of a one per rank) -- not just the configured world."""
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(round(float(one.item())))


def dry_run(args, rank, world):
    """The N > 1 plumbing without a GPU: rendezvous, sharding, the counter reductions and the JSON line, with stand-in
    per-frame numbers (frame k 'yields' 40000 keypoints, a step 'takes' 1 ms on rank 0 and a little longer on others)."""
    import torch
    import torch.distributed as dist
    import cef_loader
    sharding = cef_loader.load_submodule("sharding")
    F = args.frames_per_step
    if world > 1 or "WORLD_SIZE" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
        seen = collective_world(dist, torch.ones(1, dtype=torch.float64))
        gathered = [None] * world
        mine = sharding.frames_for_rank(F, rank, world)
        dist.all_gather_object(gathered, mine)
        d = dist
    else:
        seen, d = 1, None
        gathered = [sharding.frames_for_rank(F, rank, world)]
    cpus = None if args.no_affinity else sharding.pin_to_rank_cpus(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    dt = 1e-3 * args.steps * (1.0 + 0.01 * rank)
    t_max, kp_step, nfr = sharding.reduce_counters(d, "cpu", dt, NFEATURES * F, F)
    per_rank_s = sharding.gather_per_rank(d, "cpu", dt)
    ncpu = sharding.gather_per_rank(d, "cpu", len(cpus) if cpus else 0)
    if rank == 0:
        frames = sorted(sum(gathered, []))
        print(json.dumps({"metric": "Mkeypoints/s detectAndCompute (8K, 40k kp, BAD512)", "dry_run": True,
                          "value": round(kp_step * args.steps / t_max / 1e6, 3), "unit": "Mkeypoints/s",
                          "n_gpus": world, "rccl_world": seen, "backend": args.backend, "steps": args.steps,
                          "warmup": args.warmup, "frames_per_step": int(nfr),
                          "frames_each_once": frames == list(range(F * world)), "scaling": "weak",
                          "per_rank": {"ms_per_frame": [round(t / args.steps / F * 1e3, 4) for t in per_rank_s],
                                       "value": [round(NFEATURES * F * args.steps / t / 1e6, 3) for t in per_rank_s],
                                       "cpus_pinned": [int(n) for n in ncpu]}}), flush=True)
    if d is not None:
        d.barrier()
        d.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 1000 steps x 8 frames x ~0.33 ms = a timed region of ~2.7 s (long enough for the driver's
    # SMI sampler to see the GPU busy, and for the clocks to be what a sustained load gets)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames-per-step", type=int, default=8)
    ap.add_argument("--streams", type=int, default=3,
                    help="independent frames in flight per GPU: one context + one HIP stream each (a context is not "
                         "re-entrant, like the reference's EfficientFeaturesImpl; frames are independent)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block (BASELINE.json C2 / C3 / C4 and the README rows, tools/bench_configs.py; N = 1 only)")
    ap.add_argument("--config-iters", type=int, default=30)
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="when the timed region of --steps is shorter than this, run the same steps for this long behind it and "
                         "report them as `sustained` (the driver runs --steps 20: 53 ms; its SMI sampler and the clocks need a "
                         "load that lasts); 0 = off")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for the counters: nccl (= RCCL, the GPU path) or gloo (only with --dry-run)")
    ap.add_argument("--no-affinity", action="store_true",
                    help="do not pin each rank to its slice of the node's CPUs (sharding.pin_to_rank_cpus; N > 1 only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 only: initialise torch.distributed with the nccl (= RCCL) backend at world size 1 and run the counter "
                         "reductions / gathers through it on device tensors, exactly as the N > 1 ranks do (VERDICT r5 item 6: no "
                         "multi-GPU box has ever been reachable, so this is the only way RCCL itself executes in a recorded run)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: run the launcher, the rendezvous, the frame sharding and the counter reductions with "
                         "stand-in per-frame numbers (CPU test of the N > 1 plumbing, tests/test_bench_launcher.py)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.backend == "gloo" and not args.dry_run:
        raise SystemExit("--backend gloo is only valid with --dry-run (the product path has no CPU fallback)")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)          # bench.py --gpus N called directly: spawn the N ranks ourselves

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if args.dry_run:
        return dry_run(args, rank, world)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes on this host driver (before HIP starts)
    import torch
    import cef_loader
    cef = cef_loader.load()
    from tools import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    if torch.cuda.device_count() < world or local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    sharding = cef_loader.load_submodule("sharding")
    cpus = None
    if world > 1 and not args.no_affinity:
        # one process per GPU: each rank keeps its host threads on its own slice of the node's cores
        cpus = sharding.pin_to_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    dist = None
    rccl_world = 1
    rccl_exercised = False
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)   # nccl == RCCL on ROCm
        rccl_exercised = True
        rccl_world = collective_world(dist, torch.ones(1, dtype=torch.float64, device="cuda"))
        if rccl_world != world:
            raise SystemExit(f"bench.py: RCCL all-reduce saw {rccl_world} ranks, expected {world}")

    F = args.frames_per_step
    # per-rank frames: global frame k = rank * F + i uses seed 1000 + k.  Generated here, on the host, BEFORE the first barrier of the
    # measurement (8 ranks x 8 frames of 33 MB take the host a few seconds and differ between ranks: nothing of it may sit
    # inside a timed or barrier-bounded region -- `setup_seconds` below is where it shows)
    t_setup = time.perf_counter()
    my_frames = sharding.frames_for_rank(F, rank, world)
    frames = [torch.from_numpy(synth.synth_frame(ROWS, COLS, seed=1000 + k)).cuda() for k in my_frames]
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    timed_region_open = [False]                       # guard: no frame is generated or uploaded while it is True
    all_frames = [my_frames]
    if dist is not None:
        all_frames = [None] * world
        dist.all_gather_object(all_frames, my_frames)       # outside the timed region: which rank took which frame

    NS = max(1, min(args.streams, F))
    dets = [cef.EfficientFeatures.create(NFEATURES, 1.2, 8, 0, 20, 15, cef.EfficientFeatures.BAD_512) for _ in range(NS)]
    streams = [torch.cuda.Stream() for _ in range(NS)]
    det = dets[0]
    kps = [torch.zeros((5, NFEATURES), dtype=torch.float32, device="cuda") for _ in range(F)]
    desc = [torch.zeros((NFEATURES, 64), dtype=torch.uint8, device="cuda") for _ in range(F)]
    cnt = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(F)]

    # one step = one batched call over the F frames (efx_detect_and_compute_batch_async: frame i on context / stream i % NS).
    # F = 8 frames do not divide over 3 streams: the frame order rotates from step to step, so that every stream gets the
    # same number of frames over NS steps (otherwise one stream finishes a third early and the tail runs on two)
    def rot(lst, r):
        return lst[r:] + lst[:r]
    batches = [cef.Batch(dets, streams, rot(frames, r), rot(kps, r), rot(desc, r), rot(cnt, r), NFEATURES) for r in range(NS)]
    step_no = [0]

    def step():
        batches[step_no[0] % NS].run()
        step_no[0] += 1

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # HIP event pairs around the kernels of one frame per step on context 0 (each pair costs a few microseconds of
    # stream idle time, so not on every frame;
    # on the first PROF_STEPS steps only: a long run needs no more samples)
    PROF_STEPS = min(args.steps, int(os.environ.get("EFX_BENCH_PROF_STEPS", "24")))
    det.profileEnable(PROF_STEPS * 12 + 12, stride=max(1, F // NS))
    # one event per step and stream (no host wait): per-step spread, reported next to the mean
    marks = [[torch.cuda.Event(enable_timing=True) for _ in range(NS)] for _ in range(args.steps + 1)]
    barrier()
    for s_, ev in zip(streams, marks[0]):
        ev.record(s_)
    n_frames_before = len(frames)
    timed_region_open[0] = True
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
        for s_, ev in zip(streams, marks[k + 1]):
            ev.record(s_)
    barrier()
    dt = time.perf_counter() - t0
    timed_region_open[0] = False
    assert len(frames) == n_frames_before, "frames were generated inside the timed region"
    ms, lvl = det.profileRead()
    # a step is over when its last stream is: time of step k = latest end of step k - latest end of step k-1
    ends = np.array([[marks[0][0].elapsed_time(marks[k][s_]) for s_ in range(NS)] for k in range(args.steps + 1)]).max(axis=1)
    step_ms = np.diff(ends)

    nkp = int(sum(int(c.item()) for c in cnt))      # keypoints per step on this rank
    # RCCL over xGMI only for the counters: MAX of the time, SUM of the keypoints
    t_max, kp_step, _ = sharding.reduce_counters(dist, "cuda", dt, nkp, F)
    kp_total = kp_step * args.steps                  # keypoints all ranks processed in the timed region
    # every rank's own time / keypoints / set-up (all-gathers of one double: counters, like the reductions): which GPU the
    # max-over-ranks time comes from, and how far the ranks are apart
    per_rank_s = sharding.gather_per_rank(dist, "cuda", dt)
    per_rank_kp = sharding.gather_per_rank(dist, "cuda", nkp)
    per_rank_setup = sharding.gather_per_rank(dist, "cuda", t_setup)
    per_rank_cpus = sharding.gather_per_rank(dist, "cuda", len(cpus) if cpus else 0)

    # the same steps again for --sustain-seconds when the K timed steps were over sooner (every rank; `value` stays the K steps)
    sustained = None
    if args.sustain_seconds > 0 and dt < args.sustain_seconds:
        for d_ in dets:
            d_.profileEnable(0)
        ks = int(np.ceil(1.15 * args.sustain_seconds / (dt / args.steps)))     # a long run is a little faster per step than a short one
        barrier()
        t0 = time.perf_counter()
        for _ in range(ks):
            step()
        barrier()
        dts = time.perf_counter() - t0
        ts_max, kps_step, _ = sharding.reduce_counters(dist, "cuda", dts, nkp, F)
        sustained = {"steps": ks, "seconds": round(ts_max, 3), "value": round(kps_step * ks / ts_max / 1e6, 3), "unit": "Mkeypoints/s",
                     "ms_per_frame": round(ts_max / ks / F * 1e3, 4),
                     "what": "the timed steps repeated behind the timed region, same barriers, max over ranks"}

    if rank == 0:
        px, bytes_frame = detect_algorithmic_bytes(det, ROWS, COLS)
        stats = dets[0].lastLevelStats()
        n_corners = float(sum(s["n_candidates"] for s in stats))      # FAST corners of the last frame of context 0
        n_surv = float(sum(s["n_after_nms"] for s in stats))
        n_kp = nkp / F
        # Algorithmic HBM bytes per launch: what any implementation of the stage must move.
        #   fast_kernel    every pyramid level read once (sum_s P_s = 3.096 B per input pixel) + 4 B per corner out
        #   harris_kernel  9x9 footprints lie inside the levels (read once more) + 4 B in / 4 B out per corner
        #   nms_kernel     8 B per corner in, 8 B per survivor out (per-cell maxima are cache traffic)
        #   bad_det_kernel (EFX_NO_LEVEL_BLUR) the (blur-extended) windows lie inside the levels: sum_s P_s + 80 B record in + 64 B out per keypoint
        #                  (the reference design -- per-level blur + global integral images, SURVEY 8d -- moves
        #                  2 F P + 5 sum P_s + gathers ~ 1.06 GB for the same stage; reported as survey_design_bytes)
        #   resize chain   level s read + level s+1 written, s = 0..6: (2 F - 2) P + P_0 - P_7 ... = sum_s P_s + sum_{s>=1} P_s - P_7
        sumP = float(sum(px))
        # the chain since round 5 (resize_rows_kernel, launches make levels (1,2) (3,4) (5,6,7): EFX_ROWS_SPLIT default): levels 0, 2, 4 are
        # read, every level >= 1 is written once -- 126 MB at 8K (measured traffic 132 MB, profiles/r05_counters.json `resize_chain`);
        # the per-level chain of rounds 1 -- 4 read every level once more: sum(px[:-1]) + sum(px[1:]) = 170 MB
        split = [int(v) for v in os.environ.get("EFX_ROWS_SPLIT", "2,2,3").split(",") if v.strip()]
        srcs, s_ = [], 0
        for n_ in split:
            if s_ >= len(px) - 1:
                break
            srcs.append(s_)
            s_ += n_
        while s_ < len(px) - 1:
            srcs.append(s_)
            s_ += split[-1] if split else 1
        chain_bytes = float(sum(px[i] for i in srcs) + sum(px[1:]))
        #   blur_levels_kernel (round 4) every level read once, its blurred copy written once: 2 sum_s P_s; the keypoints are then
        #                  described a wave each on the blurred levels (bad_raw_kernel: windows inside the levels + record + descriptor)
        level_blur = bool((lvl == 11).any())
        # round 6: fast_kernel leaves 2 B per corner (16-bit tile coordinates in the tile's slot), harris_kernel reads them and
        # writes one 8-B record per corner below the level's cap
        kinfo = {0: ("fast_kernel", sumP + 2 * n_corners), 1: ("harris_kernel", sumP + 10 * n_corners),
                 2: ("nms_kernel", 8 * n_corners + 8 * n_surv),
                 10: ("bad_raw_kernel" if level_blur else "bad_det_kernel", sumP + 144 * n_kp), 11: ("blur_levels_kernel", 2 * sumP)}

        def table(ms_, lvl_, frames_per_launch=1):
            t = {}
            for code, (name, nbytes) in kinfo.items():
                m = ms_[lvl_ == code]
                if len(m):
                    avg = float(m.mean())
                    nbytes = nbytes * frames_per_launch        # a frame-batched launch (round 6) moves every frame's bytes
                    t[name] = {"avg_launch_ms": round(avg, 5), "launches_timed": int(len(m)), "algorithmic_bytes": nbytes,
                               "achieved": round(nbytes / (avg * 1e-3) / 1e9, 1), "frac": round(nbytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            return t

        def chain_ms_per_frame(ms_, lvl_):
            nl = max(1, int((lvl_ == 0).sum()))
            return float(ms_[lvl_ >= 100].sum()) / nl            # codes 100 + s: the resize launches

        # context 0 owns the frames 0, NS, 2 NS, ... of a step: they go through ONE launch of every kernel (round 6)
        fpl0 = len(range(0, F, NS))
        live = table(ms, lvl, fpl0)
        live_chain = chain_ms_per_frame(ms, lvl) / fpl0
        # the same kernels with nothing else running (frames on ONE stream, after the timed region): a kernel property,
        # unlike the live durations, which stretch with the number of frames sharing the GPU -- this is what `frac` quotes
        det.profileEnable(512, stride=1)
        for i in range(min(8, F)):
            det.detectAndComputeAsync(frames[i], kps[i], desc[i], cnt[i], capacity=NFEATURES)
        torch.cuda.synchronize()
        ms2, lvl2 = det.profileRead()
        iso = table(ms2, lvl2)
        iso_chain = chain_ms_per_frame(ms2, lvl2)
        dom = max(iso, key=lambda k: iso[k]["avg_launch_ms"]) if iso else None

        # PMC-derived figures (traffic, VALU, LDS) are not measured in this run: they come from the newest committed
        # profiles/rNN_counters.json -- and only when that file was collected from THIS tree's kernel sources (its
        # source_sha256 stamp, tools/counters_json.py); otherwise they are null with the reason
        import glob
        from tools import counters_json as cj
        cfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_counters.json")))
        prof_all = json.load(open(cfiles[-1])) if cfiles else {}
        stamp, here = prof_all.get("source_sha256"), cj.source_digests(ROOT)
        counters_ok = bool(prof_all) and stamp == here
        prof = prof_all if counters_ok else {}
        counters_info = {"file": os.path.relpath(cfiles[-1], ROOT) if cfiles else None, "commit": prof_all.get("git_head"),
                         "sources_match_this_tree": counters_ok}
        if not counters_ok:
            counters_info["why_null"] = ("no counters file" if not prof_all else "no source stamp in the counters file" if not stamp else
                                         "kernel sources changed since the counters were collected: " +
                                         ", ".join(k for k in here if (stamp or {}).get(k) != here[k]))
        roof = {"bound": "hbm", "bound_that_binds": "valu (instruction issue; see roofline.valu, which carries the fraction against the guide's "
                                                    "2-cycle peak AND against the measured half-rate / own-mix ceilings, DESIGN.md section 9)",
                "counters": counters_info, "kernel": dom, "achieved": iso[dom]["achieved"] if dom else None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": iso[dom]["frac"] if dom else None,
                "traffic": (prof.get("traffic_bytes_per_launch", {}) or {}).get(dom),
                "algorithmic_bytes_per_launch": iso[dom]["algorithmic_bytes"] if dom else None,
                "avg_launch_ms": iso[dom]["avg_launch_ms"] if dom else None,
                "launches_timed": iso[dom]["launches_timed"] if dom else 0,
                "how": "dominant kernel = longest average launch; duration from HIP event pairs recorded by the library on the "
                       "launch stream around that kernel, frames on one stream (after the timed region), so that the figure is a "
                       "property of the kernel; `live` = the same pairs inside the timed region, three frames in flight",
                "kernels_isolated": iso,
                "live": {"concurrent_streams": NS, "frames_per_launch": fpl0, "kernels": live, "resize_chain_ms_per_frame": round(live_chain, 5)},
                "whole_frame": {"survey_8d_bytes_per_frame": 0.9e9, "ms_per_frame": round(t_max / args.steps / F * 1e3, 4),
                                "achieved": round(0.9e9 / (t_max / args.steps / F) / 1e9, 1),
                                "frac": round(0.9e9 / (t_max / args.steps / F) / 1e9 / HBM_PEAK_GBS, 4)}}
        if dom and dom.startswith("bad"):
            roof["survey_design_bytes"] = 2 * sumP + 5 * sumP + 338e6 + 80 * n_kp
        # the pass the north star names: pyramid (resize chain) + FAST, bytes = levels read by FAST + levels read / written by
        # the chain + corners out, over the isolated durations
        if "fast_kernel" in iso:
            pf_ms = iso_chain + iso["fast_kernel"]["avg_launch_ms"]
            pf_bytes = chain_bytes + iso["fast_kernel"]["algorithmic_bytes"]
            roof["pyramid_fast"] = {"algorithmic_bytes": pf_bytes, "resize_chain_bytes": chain_bytes, "resize_chain_source_levels": srcs,
                                    "resize_chain_note": "bytes of the row-walking chain under the default EFX_ROWS_SPLIT (an aligned 8K frame takes "
                                                         "it; a caller's unaligned / odd-width image falls back to the tiled per-level chain, which "
                                                         "reads every level once more -- ADVICE r5)",
                                    "resize_chain_ms": round(iso_chain, 5),
                                    "fast_kernel_ms": iso["fast_kernel"]["avg_launch_ms"], "ms": round(pf_ms, 5),
                                    "achieved": round(pf_bytes / (pf_ms * 1e-3) / 1e9, 1),
                                    "frac": round(pf_bytes / (pf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "survey_8d_bytes": float(sum(px) + sum(px[1:])),
                                    "frac_vs_survey_8d_bytes": round((sum(px) + sum(px[1:])) / (pf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # the bound that binds: VALU instruction issue (integral / box phases of the BAD kernel: the LDS pipe).  VALU
        # wave-instructions per launch come from the committed SQ counter pass, the issue peak from the committed
        # micro-benchmark (profiles/valu_rate.txt): 1024 SIMDs x clock / cycles per wave-instruction
        if prof.get("valu_wave_instr_per_launch") and prof.get("issue_peak_wave_instr_per_s"):
            # three ceilings (profiles/rNN_valu_rate.txt has the per-instruction table they come from):
            #   guide      1024 SIMDs x 2.4 GHz / 2 cycles (MI355X_MICROARCH.md: wave64 VALU = 2 cycles) -- EVERY instruction full rate
            #   half_rate  measured: 4.19 cycles -- a kernel made only of half-rate instructions (packed 16-bit, perm, dot, cvt,
            #              min / max, 3-operand integer, packed fp32 ...: most of what these kernels issue)
            #   mix        the kernel's own static mix of full- and half-rate instructions (tools/valu_mix.py)
            clk = float(prof.get("clock_ghz", 2.4)) * 1e9
            peak_guide = float(prof.get("issue_peak_guide", 1024 * 2.4e9 / 2))
            peak_half = float(prof.get("issue_peak_half_rate", prof["issue_peak_wave_instr_per_s"]))
            mixc = prof.get("mix_cycles_per_wave_instr", {}) or {}

            def vrow(name, v, ms_):
                rate = v / (ms_ * 1e-3)
                row = {"valu_wave_instr": v, "avg_launch_ms": round(ms_, 5), "achieved": round(rate / 1e9, 1),
                       "frac_vs_guide_peak": round(rate / peak_guide, 4), "frac_vs_half_rate_ceiling": round(rate / peak_half, 4)}
                if mixc.get(name):
                    row["mix_cycles_per_wave_instr"] = mixc[name]
                    row["frac_vs_mix_ceiling"] = round(rate / (1024 * clk / mixc[name]), 4)
                row["frac"] = row.get("frac_vs_mix_ceiling", row["frac_vs_half_rate_ceiling"])
                return row
            vt = {}
            for name, row in iso.items():
                v = prof["valu_wave_instr_per_launch"].get(name)
                if v:
                    vt[name] = vrow(name, v, row["avg_launch_ms"])
            vchain = prof["valu_wave_instr_per_launch"].get("resize_chain")
            if vchain and iso_chain > 0:
                vt["resize_chain"] = vrow("resize_chain", vchain, iso_chain)
            tot = prof.get("valu_wave_instr_per_frame")
            roof["valu"] = {"bound": "valu", "unit": "G wave-instructions/s",
                            "peaks": {"guide": round(peak_guide / 1e9, 1), "full_rate_measured": round(float(prof.get("issue_peak_full_rate", 0)) / 1e9, 1),
                                      "half_rate_measured": round(peak_half / 1e9, 1)},
                            "peak": round(peak_guide / 1e9, 1),
                            "peak_from": "guide = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md); the measured ceilings "
                                         "are profiles/%s_valu_rate.txt: %s cycles for the full-rate instructions (32-bit add / sub / logic / right "
                                         "shift / move, fp32 add / mul / fma), %s cycles for everything else, clock %s GHz measured under load; "
                                         "`frac` = against the kernel's own static mix of the two (%s)" % (
                                             os.path.basename(cfiles[-1])[:3], prof.get("cycles_full_rate"), prof.get("cycles_half_rate"),
                                             prof.get("clock_ghz"), prof.get("mix_from")),
                            "kernels_isolated": vt}
            if tot:
                rate = tot / (t_max / args.steps / F)
                # the frame's mix: the kernels' mixes weighted by their instruction counts
                wsum = sum(prof["valu_wave_instr_per_launch"].get(k, 0) for k in mixc if k in prof["valu_wave_instr_per_launch"] and k != "resize_stream_kernel")
                fmix = (sum(prof["valu_wave_instr_per_launch"][k] * mixc[k] for k in mixc if k in prof["valu_wave_instr_per_launch"] and k != "resize_stream_kernel") / wsum) if wsum else None
                roof["valu"]["whole_frame"] = {"valu_wave_instr_per_frame": tot, "ms_per_frame": round(t_max / args.steps / F * 1e3, 4),
                                               "achieved": round(rate / 1e9, 1), "frac_vs_guide_peak": round(rate / peak_guide, 4),
                                               "frac_vs_half_rate_ceiling": round(rate / peak_half, 4)}
                if fmix:
                    roof["valu"]["whole_frame"]["mix_cycles_per_wave_instr"] = round(fmix, 3)
                    roof["valu"]["whole_frame"]["frac_vs_mix_ceiling"] = round(rate / (1024 * clk / fmix), 4)
                roof["valu"]["whole_frame"]["frac"] = roof["valu"]["whole_frame"].get("frac_vs_mix_ceiling", roof["valu"]["whole_frame"]["frac_vs_half_rate_ceiling"])
            if prof.get("lds_cycles_per_launch", {}).get(dom):
                lc = prof["lds_cycles_per_launch"][dom]
                roof["lds"] = {"kernel": dom, "lds_array_cycles_per_launch": lc, "cus": 256,
                               "busy_frac": round(lc / 256 / (prof.get("clock_ghz", 2.4) * 1e9) / (iso[dom]["avg_launch_ms"] * 1e-3), 4)}

        out = {"metric": "Mkeypoints/s detectAndCompute (8K, 40k kp, BAD512)",
               "value": round(kp_total / t_max / 1e6, 3), "unit": "Mkeypoints/s", "n_gpus": world, "rccl_world": rccl_world, "rccl_exercised": rccl_exercised,
               "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(t_max / args.steps * 1e3, 4),
               "ms_per_step_min_median_max": [round(float(step_ms.min()), 4), round(float(np.median(step_ms)), 4), round(float(step_ms.max()), 4)],
               "ms_per_frame": round(t_max / args.steps / F * 1e3, 4),
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": round(kp_total / t_max / 1e6 / BASELINE_MKPS, 2), "dtype": "u8",
               "data": "synthetic",
               "config": {"workload": "detectAndCompute BAD512 on 8K (7680x4320) synthetic frames, nfeatures=40000, "
                                      "8 levels, scale 1.2, FAST threshold 20, NMS radius 15 (BASELINE.json configs[4])",
                          "frames_per_step_per_gpu": F, "frames_per_step": F * world,
                          "frames_each_once": sorted(sum(all_frames, [])) == list(range(F * world)), "streams_per_gpu": NS,
                          "frames_per_launch": [len(range(j, F, NS)) for j in range(NS)],
                          "keypoints_per_frame": round(nkp / F, 1), "parallelism": f"frames sharded over {world} GPU(s)"},
               "sustained": sustained if sustained else {"note": "the timed region itself lasted %.2f s" % t_max},
               "per_rank": {"ms_per_frame": [round(t / args.steps / F * 1e3, 4) for t in per_rank_s],
                            "value": [round(k * args.steps / t / 1e6, 3) for k, t in zip(per_rank_kp, per_rank_s)],
                            "slowest_over_fastest": round(max(per_rank_s) / min(per_rank_s), 4),
                            "setup_seconds": [round(t, 2) for t in per_rank_setup],
                            "cpus_pinned": [int(n) for n in per_rank_cpus],
                            "note": "rank r = GPU r; `value` (top level) = all ranks' keypoints / the slowest rank's time; the set-up "
                                    "(synthetic frames generated and uploaded) lies before the first barrier"},
               "roofline": roof}
        tag = os.path.basename(cfiles[-1])[:3] if cfiles else "rNN"
        roof["profiles"] = {"frac / kernels_isolated (one stream)": "profiles/%s_kernel_stats.csv" % tag,
                            "live (the default command, three frames in flight)": "profiles/%s_kernel_stats_3streams.csv" % tag,
                            "traffic, valu, lds": "profiles/%s_counters.json <- profiles/%s_pmc_sq.txt, %s_pmc_lds.txt, %s_traffic.json, %s_valu_rate.txt, %s_valu_mix.json" % (tag, tag, tag, tag, tag, tag)}
        if not counters_ok:
            roof["valu"] = None; roof["lds"] = None

        # The reference's own protocol (samples/sample_benchmark.cpp:39-52: 1 warm-up, then N x {detectAndComputeAsync;
        # stream.waitForCompletion()}): one frame at a time on one stream, host wait included.  This is the figure that
        # corresponds cell for cell to BASELINE.md's "8.2 ms"; `value` above keeps several frames in flight.
        # "One context, one stream": the headline's three contexts, their streams and side streams are released first.  (Measured,
        # tools/microbench/lat_probe.py: with them alive every one-call figure of the process reads ~25 us higher -- 0.426 instead of
        # 0.400 ms for 8K BAD512 -- and falls back when they are gone: the waits and the per-call stream queries see every stream.)
        import gc
        del batches[:], dets[:], streams[:]
        det = None
        gc.collect()
        torch.cuda.synchronize()
        det = cef.EfficientFeatures.create(NFEATURES, 1.2, 8, 0, 20, 15, cef.EfficientFeatures.BAD_512)
        for _ in range(3):                               # a new context: first-use allocations, side stream, the per-call fork decision
            det.detectAndComputeAsync(frames[0], kps[0], desc[0], cnt[0], capacity=NFEATURES)
            torch.cuda.synchronize()
        nlat = 24
        t1 = time.perf_counter()
        for i in range(nlat):
            det.detectAndComputeAsync(frames[i % F], kps[i % F], desc[i % F], cnt[i % F], capacity=NFEATURES)
            torch.cuda.current_stream().synchronize()
        t_lat = (time.perf_counter() - t1) / nlat
        out["latency"] = {"protocol": "sample_benchmark.cpp perf(): warm-up + 24 x (detectAndComputeAsync + stream wait), one context, one stream, the step's 8 frames in turn",
                          "ms_per_frame": round(t_lat * 1e3, 4), "vs_baseline_ms": round(BASELINE_MS / (t_lat * 1e3), 2)}

        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle
            img = frames[0].cpu().numpy()
            pyoracle.set_threads(1)
            t_runs = []
            for _ in range(3):                            # three repeats of the single-thread port (~2.4 s each): min and median
                t1 = time.perf_counter()
                ref = pyoracle.detect_and_compute(img, nfeatures=NFEATURES, desc_type=pyoracle.BAD_512)
                t_runs.append(time.perf_counter() - t1)
            t_cpu = min(t_runs)
            # the same frame with OpenMP over rows / candidates / keypoints on the host's cores (bounded at 64 threads)
            nthr = max(1, min(os.cpu_count() or 1, 64))
            pyoracle.set_threads(nthr)
            pyoracle.detect_and_compute(img[:512, :512].copy(), nfeatures=100, desc_type=pyoracle.BAD_512)   # start the thread pool
            t_mts = []
            for _ in range(3):
                t1 = time.perf_counter()
                ref_mt = pyoracle.detect_and_compute(img, nfeatures=NFEATURES, desc_type=pyoracle.BAD_512)
                t_mts.append(time.perf_counter() - t1)
            t_mt = min(t_mts)
            pyoracle.set_threads(1)
            out["cpu_baseline"] = {"value": round(ref["n"] / t_cpu / 1e6, 5), "unit": "Mkeypoints/s", "cores": 1,
                                   "kind": "port", "ms_per_frame": round(t_cpu * 1e3, 1),
                                   "ms_per_frame_min_median_max": [round(min(t_runs) * 1e3, 1), round(float(np.median(t_runs)) * 1e3, 1), round(max(t_runs) * 1e3, 1)],
                                   "repeats": len(t_runs),
                                   "sample": "one 8K frame (seed 1000) of the same workload, oracle/efx_oracle.c, "
                                             f"single thread as the reference CPU module, best of {len(t_runs)}; {ref['n']} keypoints",
                                   "host_cores_available": os.cpu_count(),
                                   "all_cores": {"value": round(ref_mt["n"] / t_mt / 1e6, 5), "cores": nthr,
                                                 "ms_per_frame": round(t_mt * 1e3, 1),
                                                 "ms_per_frame_min_median_max": [round(min(t_mts) * 1e3, 1), round(float(np.median(t_mts)) * 1e3, 1), round(max(t_mts) * 1e3, 1)],
                                                 "same_result": bool(np.array_equal(ref_mt["desc"], ref["desc"]))}}
            # the bench frame doubles as a full-size parity check (bit-exact keypoints + BAD512 bytes)
            n0 = int(cnt[0].item())
            same = (n0 == ref["n"] and np.array_equal(kps[0][:, :n0].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))
                    and np.array_equal(desc[0][:n0].cpu().numpy(), ref["desc"]))
            out["parity_8k_frame0"] = bool(same)
        if world == 1 and not args.no_configs:
            # the other BASELINE.json configurations and the README's rows, reference protocol, in the driver-run line
            det = None                                    # (the headline's contexts were released before the latency rows)
            gc.collect()
            from tools import bench_configs
            out["configs"] = bench_configs.measure(cef, iters=args.config_iters, cpu_baseline=not args.no_cpu_baseline)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()                  # rank 0 is still measuring its isolated kernels / latency: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
