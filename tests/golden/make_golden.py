#!/usr/bin/env python3
"""Regenerates the golden vectors in this directory (run from the repo root: python tests/golden/make_golden.py).

What a fixture is here: the seeded input (tools/synth.py, tests/lcg_probe.py) and the EXPECTED OUTPUTS are
both stored, so the vectors do not depend on a numpy version.  Two kinds:

 * descriptors_probe.npz -- BAD256/512 and HashSIFT256/512 bytes of the SURVEY.md Appendix B probe input (LCG
   640x480 image, 200 keypoints).  These bytes hash (FNV-1a-32) to the values the survey recorded from the
   REFERENCE CPU code (modules/efficient_features/src/bad.cpp, hash_sift.cpp) run in the build container, so they
   are reference outputs; the generator refuses to write them if the hashes do not match.
 * detector_*.npz -- keypoint matrices / descriptors / intermediate stages of the detector for small seeded
   frames.  The reference has no CPU detector and cannot be built here (CUDA + OpenCV-CUDA), so these are outputs
   of the spec-defining CPU restatement (oracle/): regression vectors for the oracle AND the HIP
   path, "parity unpinned" against the reference itself.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O          # noqa: E402
from tests.lcg_probe import REFERENCE_HASHES, fnv1a32, probe_input   # noqa: E402
from tools import synth                   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# (name, generator, rows, cols, seed, detector kwargs)
DETECTOR_CASES = [
    ("synth_240x320", "synth", 240, 320, 31, dict(nfeatures=1500)),
    ("synth_480x640", "synth", 480, 640, 1000, dict(nfeatures=4000)),
    ("synth_480x640_r5_t40", "synth", 480, 640, 32, dict(nfeatures=4000, nonmax_radius=5, fast_threshold=40)),
    ("noise_200x260_cap", "noise", 200, 260, 7, dict(nfeatures=3000)),
    # bright squares on a grid of period `seed`: thousands of corners with identical Harris responses, which suppress
    # each other (the <= rule of IsMaxPoint, spec S3): level 0 has 6583 FAST corners and no survivor
    ("squares_200x260_ties", "squares", 200, 260, 12, dict(nfeatures=2000)),
]


def frame(kind, rows, cols, seed):
    if kind == "squares":
        y, x = np.mgrid[0:rows, 0:cols]
        return ((((x % seed) < seed // 2) & ((y % seed) < seed // 2)) * 200).astype(np.uint8)
    return synth.synth_frame(rows, cols, seed=seed) if kind == "synth" else synth.noise_frame(rows, cols, seed=seed)


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None        # regenerate a single detector fixture by name
    img, kps = probe_input()
    out = {}
    for nbits in (256, 512):
        out[f"bad{nbits}"] = O.bad_compute(img, kps, nbits)
        out[f"hashsift{nbits}"] = O.hashsift_compute(img, kps, nbits)
        for kind in ("bad", "hashsift"):
            h = fnv1a32(out[f"{kind}{nbits}"])
            if h != REFERENCE_HASHES[(kind, nbits)]:
                raise SystemExit(f"{kind}{nbits}: hash {h:08x} != reference {REFERENCE_HASHES[(kind, nbits)]:08x}")
    out["image"] = img; out["keypoints"] = kps
    if only is None:
        np.savez_compressed(os.path.join(HERE, "descriptors_probe.npz"), **out)

    for name, kind, rows, cols, seed, kw in DETECTOR_CASES:
        if only is not None and name != only:
            continue
        im = frame(kind, rows, cols, seed)
        d = {"image": im}
        for dt, tag in ((O.BAD_256, "bad256"), (O.BAD_512, "bad512"), (O.HASH_SIFT_256, "hashsift256"), (O.HASH_SIFT_512, "hashsift512")):
            r = O.detect_and_compute(im, desc_type=dt, **kw)
            if "kps" not in d:
                d["kps"] = r["kps"].view(np.uint32)              # raw bits of the 5xN matrix
                d["lvl_xy"] = r["lvl_xy"]
                d["n_candidates"] = np.array(r["stats"]["n_candidates"]); d["n_after_cap"] = np.array(r["stats"]["n_after_cap"])
                d["n_after_nms"] = np.array(r["stats"]["n_after_nms"]); d["n_kept"] = np.array(r["stats"]["n_kept"])
            d[tag] = r["desc"]
        # intermediate stages (level 1 of the pyramid, its blur, FAST corners of level 0)
        d["level1"] = O.pyramid_level(im, 1)
        d["level1_blur"] = O.gaussian7(d["level1"])
        d["fast_l0"] = O.fast9_detect(im, threshold=kw.get("fast_threshold", 20), border=15)
        np.savez_compressed(os.path.join(HERE, f"detector_{name}.npz"), **d)
        print(name, "keypoints", d["kps"].shape[1], "fast corners L0", len(d["fast_l0"]))


if __name__ == "__main__":
    main()
