#!/usr/bin/env python3
"""First-frame stress: P fresh processes at a time, each creating ONE context and running ONE frame
(tools/microbench/first_frame.cpp), R rounds; every result is compared with the oracle's digest.  This is the one case the
process-wide block cache cannot cover (a process's first context always runs on freshly mapped memory).
  python tools/microbench/first_frame_stress.py --procs 16 --rounds 100 [--rows 420 --cols 520]
Prints one JSON line: frames run, results that differ, processes that died (exit code / signal)."""
import argparse
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from oracle import pyoracle
from tools import synth

FNV0, FNVP, M64 = 1469598103934665603, 1099511628211, (1 << 64) - 1


def fnv(data, h):
    for b in data:
        h = ((h ^ b) * FNVP) & M64
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--rounds", type=int, default=50)
    ap.add_argument("--rows", type=int, default=420)
    ap.add_argument("--cols", type=int, default=520)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--background", type=int, default=0, help="processes that keep one context busy with frames all the while (oversubscription)")
    args = ap.parse_args()
    exe = os.path.join("tools", "microbench", "first_frame")
    img = synth.synth_frame(args.rows, args.cols, seed=77)
    raw = os.path.join("gpurun_out", "first_frame.raw")
    os.makedirs("gpurun_out", exist_ok=True)
    img.tofile(raw)
    ref = pyoracle.detect_and_compute(img, nfeatures=args.nfeatures, desc_type=pyoracle.BAD_256)
    n = ref["n"]
    h = fnv(np.int32(n).tobytes(), FNV0)
    k = ref["kps"].view(np.uint32)
    for r in range(5):
        h = fnv(k[r, :n].tobytes(), h)
    h = fnv(ref["desc"][:n].tobytes(), h)
    want = "%016x" % h
    base = [exe, raw, str(args.rows), str(args.cols), str(args.nfeatures)]
    one = subprocess.run(base + ["0"], capture_output=True, text=True)
    assert one.returncode == 0 and one.stdout.split()[0] == want, ("a single process does not reproduce the oracle", one.stdout, one.stderr, want)
    res = {}
    for label, env in (("block_cache", {}), ("EFX_NO_BLOCK_CACHE", {"EFX_NO_BLOCK_CACHE": "1"})):
        differ = died = 0
        t0 = time.time()
        bg = [subprocess.Popen(base + [want, "100000000"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, **env))
              for _ in range(args.background)]
        time.sleep(3.0 if bg else 0.0)
        for _ in range(args.rounds):
            ps = [subprocess.Popen(base + [want], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env)) for _ in range(args.procs)]
            for p in ps:
                p.communicate()
                if p.returncode == 3:
                    differ += 1
                elif p.returncode != 0:
                    died += 1
        bg_dead = sum(1 for p in bg if p.poll() is not None)
        for p in bg:
            p.kill()                       # exact PIDs this script started
            p.wait()
        res[label] = {"first_frames": args.rounds * args.procs, "differ": differ, "died": died, "background": args.background,
                      "background_died": bg_dead, "seconds": round(time.time() - t0, 1)}
    print(json.dumps({"frame": [args.rows, args.cols], "keypoints": int(n), "procs_at_a_time": args.procs, "results": res}))


if __name__ == "__main__":
    main()
