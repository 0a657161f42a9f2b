#!/usr/bin/env python3
"""Pins the detector's two constant tables on data the REFERENCE holds.

Run in the build container (reads /root/reference BY PATH, like tools/extract_params.py; nothing of the reference's text
is copied into the repo -- only SHA-256 digests and lengths are written):

  c_table  modules/cuda_efficient_features/src/cuda_fast.cu:31   8129 bytes; isKeyPoint (:160-166) looks a 16-bit ring
           mask m with popcount > 8 up as  c_table[(m >> 3) - 63] & (1 << (m & 7))
  U_MAX    modules/cuda_efficient_features/src/cuda_efficient_features.cu:143   17 ints; IC_Angle's row half-widths

and the four learned descriptor parameter tables, parsed here with this file's own parser (not
tools/extract_params.py's) and hashed in the canonical layout of params/*.bin:

  bad{256,512}       modules/efficient_features/src/bad.p256.h:27,94 / bad.p512.h:209,340
                     int32[N][5] {x1, x2, y1, y2, radius} then float32[N] thresholds (double literal -> float)
  hashsift{256,512}  modules/efficient_features/src/hash_sift.p{256,512}.h   float64[N][129]

tests/test_reference_table_pins.py checks params/*.bin (and, on the GPU box, the blobs embedded in libefx_hip.so) against
these digests.  It regenerates the two detector tables from the oracle's own predicates and compares the digests; the GPU
test drives fast_kernel with ring patterns for all 65 536 masks x both polarities against the same predicate.
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference/modules/cuda_efficient_features/src"
REF_CPU = "/root/reference/modules/efficient_features/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_table_pins.json")


def parse_array(path, name):
    text = open(path).read()
    m = re.search(r"\b" + re.escape(name) + r"\s*\[\s*\]\s*=\s*\{([^}]*)\}", text)
    if not m:
        raise SystemExit(f"{name} not found in {path}")
    line = text[:m.start()].count("\n") + 1
    vals = [int(t, 0) for t in re.findall(r"0[xX][0-9a-fA-F]+|\d+", m.group(1))]
    return vals, line


def c_table_bytes_from_predicate(has_arc9):
    """The table isKeyPoint's index formula implies: byte i holds masks 8 (i + 63) .. 8 (i + 63) + 7, bit = mask & 7."""
    t = np.zeros(8129, dtype=np.uint8)
    for mask in range(63 * 8, 65536):
        if has_arc9(mask):
            t[(mask >> 3) - 63] |= 1 << (mask & 7)
    return t


def c_tokens(path):
    """The header as a flat token stream with comments removed (numbers, identifiers, punctuation)."""
    text = open(path, encoding="utf-8", errors="replace").read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return text


def numbers_of(path, name):
    """All numeric literals of the initialiser of array `name`, in order, and the line the array starts at."""
    raw = open(path, encoding="utf-8", errors="replace").read()
    text = c_tokens(path)
    i = text.find(name)
    if i < 0:
        raise SystemExit(f"{name} not found in {path}")
    i = text.index("=", i)
    depth, j = 0, text.index("{", i)
    k = j
    while True:
        ch = text[k]
        depth += ch == "{"
        depth -= ch == "}"
        if depth == 0:
            break
        k += 1
    body = text[j:k + 1]
    nums = re.findall(r"[-+]?(?:\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+|\d+)", body)
    line = raw[:raw.find(name)].count("\n") + 1
    return nums, line


def pin_descriptor_tables(pins):
    for n in (256, 512):
        path = os.path.join(REF_CPU, f"bad.p{n}.h")
        boxes, l1 = numbers_of(path, f"box_pair_params_{n}")
        thr, l2 = numbers_of(path, f"thresholds_{n}")
        assert len(boxes) == 5 * n and len(thr) == n, (len(boxes), len(thr))
        blob = np.array([int(v) for v in boxes], dtype="<i4").tobytes() + \
            np.array([float(v) for v in thr], dtype=np.float64).astype("<f4").tobytes()
        # the copy the CUDA module compiles is the same table
        b2, _ = numbers_of(os.path.join(REF, f"bad.p{n}.h"), f"box_pair_params_{n}")
        t2, _ = numbers_of(os.path.join(REF, f"bad.p{n}.h"), f"thresholds_{n}")
        assert b2 == boxes and t2 == thr
        pins[f"bad{n}"] = {"source": f"modules/efficient_features/src/bad.p{n}.h:{l1},{l2}", "bytes": len(blob),
                           "layout": "int32[N][5] {x1,x2,y1,y2,radius} + float32[N] thresholds, little endian",
                           "sha256": hashlib.sha256(blob).hexdigest()}
        path = os.path.join(REF_CPU, f"hash_sift.p{n}.h")
        vals, l3 = numbers_of(path, f"HASH_SIFT_{n}_VALS")
        assert len(vals) == 129 * n, len(vals)
        v2, _ = numbers_of(os.path.join(REF, f"hash_sift.p{n}.h"), f"HASH_SIFT_{n}_VALS")
        assert v2 == vals
        blob = np.array([float(v) for v in vals], dtype="<f8").tobytes()
        pins[f"hashsift{n}"] = {"source": f"modules/efficient_features/src/hash_sift.p{n}.h:{l3}", "bytes": len(blob),
                                "layout": "float64[N][129] row-major, little endian", "sha256": hashlib.sha256(blob).hexdigest()}


def main():
    ctab, l1 = parse_array(os.path.join(REF, "cuda_fast.cu"), "c_table")
    umax, l2 = parse_array(os.path.join(REF, "cuda_efficient_features.cu"), "U_MAX")
    ctab = np.array(ctab, dtype=np.uint8)
    umax = np.array(umax, dtype="<i4")
    # sanity: the reference's table is exactly ">= 9 circularly contiguous set bits" (bits of masks with popcount <= 8 are 0)
    def arc9(m):
        d = m | (m << 16)
        d &= d >> 1; d &= d >> 2; d &= d >> 4; d &= d >> 1
        return (d & 0xffff) != 0
    regen = c_table_bytes_from_predicate(arc9)
    pins = {"note": "SHA-256 of tables held by the reference; written by tools/pin_reference_tables.py in the build container",
            "c_table": {"source": "modules/cuda_efficient_features/src/cuda_fast.cu:%d" % l1, "len": int(ctab.size),
                        "sha256": hashlib.sha256(ctab.tobytes()).hexdigest(),
                        "equals_arc9_predicate": bool(np.array_equal(ctab, regen))},
            "U_MAX": {"source": "modules/cuda_efficient_features/src/cuda_efficient_features.cu:%d" % l2, "len": int(umax.size),
                      "dtype": "int32 little endian", "sha256": hashlib.sha256(umax.tobytes()).hexdigest()}}
    pin_descriptor_tables(pins)
    json.dump(pins, open(OUT, "w"), indent=1)
    print(json.dumps(pins, indent=1))


if __name__ == "__main__":
    sys.exit(main())
