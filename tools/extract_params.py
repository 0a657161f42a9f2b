#!/usr/bin/env python3
"""Extract the learned descriptor parameters from the reference headers into binary blobs.

Runs ONLY in the build container (needs /root/reference).  The blobs are data, not code:
  * BAD box pairs + thresholds   <- modules/efficient_features/src/bad.p256.h:27,94
                                    modules/efficient_features/src/bad.p512.h:209,340
  * HashSIFT projection matrices <- modules/efficient_features/src/hash_sift.p{256,512}.h:22
(the copies under modules/cuda_efficient_features/src are byte-identical).  The learned values are
(c) the BAD / HashSIFT authors and Fixstars, Apache-2.0; see params/NOTICE.

Blob layouts (little endian):
  bad{N}.bin      : int32[N][5] {x1, x2, y1, y2, boxRadius} then float32[N] thresholds
  hashsift{N}.bin : float64[N][129] row-major (column 0 multiplies the constant-1 bias element);
                    converted to float32 at load time exactly like the reference's
                    Mat(nbits,129,CV_64F).convertTo(CV_32F)  (hash_sift.cpp:390-392).
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference/modules/efficient_features/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-efficient-features_amd", "params")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return text


def array_body(text, name):
    m = re.search(re.escape(name) + r"\s*\[\s*\]\s*=\s*\{(.*?)\}\s*;", text, flags=re.S)
    if not m:
        raise SystemExit(f"array {name} not found")
    return m.group(1)


def extract_bad(nbits):
    path = os.path.join(REF, f"bad.p{nbits}.h")
    raw = open(path, "rb").read()
    text = strip_comments(raw.decode("utf-8", "replace"))
    boxes_txt = array_body(text, f"box_pair_params_{nbits}")
    boxes = np.array([[int(v) for v in grp.split(",")] for grp in re.findall(r"\{([^{}]*)\}", boxes_txt)], dtype=np.int32)
    assert boxes.shape == (nbits, 5), boxes.shape
    thr_txt = array_body(text, f"thresholds_{nbits}")
    # C semantics: a double literal converted to float (round to nearest)
    thr = np.array([float(v) for v in thr_txt.replace("\n", " ").split(",") if v.strip()], dtype=np.float64).astype(np.float32)
    assert thr.shape == (nbits,), thr.shape
    # every box lies inside the 32x32 patch
    assert (boxes[:, :4] - boxes[:, 4:5] >= 0).all() and (boxes[:, :4] + boxes[:, 4:5] <= 31).all()
    blob = boxes.tobytes() + thr.tobytes()
    return blob, hashlib.md5(raw).hexdigest()


def extract_hashsift(nbits):
    path = os.path.join(REF, f"hash_sift.p{nbits}.h")
    raw = open(path, "rb").read()
    text = strip_comments(raw.decode("utf-8", "replace"))
    body = array_body(text, f"HASH_SIFT_{nbits}_VALS")
    vals = np.array([float(v) for v in body.split(",") if v.strip()], dtype=np.float64)
    assert vals.size == nbits * 129, vals.size
    return vals.tobytes(), hashlib.md5(raw).hexdigest()


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    for nbits in (256, 512):
        blob, md5 = extract_bad(nbits)
        fn = f"bad{nbits}.bin"
        open(os.path.join(OUT, fn), "wb").write(blob)
        manifest[fn] = {"source": f"modules/efficient_features/src/bad.p{nbits}.h", "source_md5": md5,
                        "blob_md5": hashlib.md5(blob).hexdigest(), "bytes": len(blob)}
        blob, md5 = extract_hashsift(nbits)
        fn = f"hashsift{nbits}.bin"
        open(os.path.join(OUT, fn), "wb").write(blob)
        manifest[fn] = {"source": f"modules/efficient_features/src/hash_sift.p{nbits}.h", "source_md5": md5,
                        "blob_md5": hashlib.md5(blob).hexdigest(), "bytes": len(blob)}
    json.dump(manifest, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(manifest, indent=1, sort_keys=True))


if __name__ == "__main__":
    sys.exit(main())
