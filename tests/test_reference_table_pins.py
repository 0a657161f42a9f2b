"""The detector's two constant tables, pinned on data the reference holds.

tests/golden/reference_table_pins.json carries SHA-256 digests of
  c_table  (modules/cuda_efficient_features/src/cuda_fast.cu:31, used by isKeyPoint :160-166) and
  U_MAX    (modules/cuda_efficient_features/src/cuda_efficient_features.cu:143),
computed from the reference's files by tools/pin_reference_tables.py in the build container.  Here both are regenerated
from the predicates the oracle / the HIP kernels actually use and must hash to the same digests."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = json.load(open(os.path.join(HERE, "golden", "reference_table_pins.json")))

# circle order of cuda_fast.cu:179-207 (bit k = position k): starts at (x, y+3) and walks towards +x
FAST_DX = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
FAST_DY = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]


def table_from_predicate(pred):
    """The table the reference's index formula implies: isKeyPoint reads c_table[(m >> 3) - 63] & (1 << (m & 7))."""
    t = np.zeros(8129, dtype=np.uint8)
    for m in range(63 * 8, 65536):
        if pred(m):
            t[(m >> 3) - 63] |= 1 << (m & 7)
    return t


def test_oracle_arc9_predicate_equals_reference_c_table(oracle):
    t = table_from_predicate(oracle.has_arc9)
    assert t.size == PINS["c_table"]["len"]
    assert hashlib.sha256(t.tobytes()).hexdigest() == PINS["c_table"]["sha256"]
    # masks the table does not cover (< 504) have at most 8 bits set: popcount test of isKeyPoint fails, predicate must too
    assert not any(oracle.has_arc9(m) for m in range(0, 504))


def test_oracle_umax_equals_reference_u_max(oracle):
    u = np.array(oracle.ic_umax(), dtype="<i4")
    assert u.size == PINS["U_MAX"]["len"]
    assert hashlib.sha256(u.tobytes()).hexdigest() == PINS["U_MAX"]["sha256"]
    # calcUMax of the HPatches exporter (hpatches_description.cpp:107-126) for a 31-px patch is the same table
    v = np.array(oracle.calc_umax(31), dtype="<i4")
    assert hashlib.sha256(v.tobytes()).hexdigest() == PINS["U_MAX"]["sha256"]


def ring_image(masks, polarity, p=100, t=20, pitch_px=24, per_row=80):
    """One 7x7 ring pattern per mask on a flat background of value p: ring pixel k is just past the threshold
    (p +- (t+1)) when bit k of the mask is set and exactly AT the threshold (p +- t, not a hit: diffType is strict,
    cuda_fast.cu:36-40) when it is clear.  Returns the image and the centre coordinates."""
    n = len(masks)
    rows_p = (n + per_row - 1) // per_row
    img = np.full((24 + rows_p * pitch_px + 24, 24 + per_row * pitch_px + 24), p, dtype=np.uint8)
    sgn = 1 if polarity == "bright" else -1
    idx = np.arange(n)
    cx = 24 + (idx % per_row) * pitch_px + 8
    cy = 24 + (idx // per_row) * pitch_px + 8
    m = np.asarray(masks)
    for k in range(16):
        bit = (m >> k) & 1
        img[cy + FAST_DY[k], cx + FAST_DX[k]] = (p + sgn * (t + bit)).astype(np.uint8)
    return img, cx, cy


@pytest.mark.gpu
@pytest.mark.parametrize("polarity", ["bright", "dark"])
def test_fast_kernel_equals_reference_c_table_for_all_masks(polarity):
    """fast_kernel on ring patterns for all 65 536 masks (x both polarities): the set of masks it calls a corner,
    written as the reference's table, hashes to the reference's c_table."""
    import torch
    import cef_loader
    cef = cef_loader.load()
    detected = np.zeros(65536, dtype=bool)
    chunk = 6400
    for m0 in range(0, 65536, chunk):
        masks = np.arange(m0, min(m0 + chunk, 65536))
        img, cx, cy = ring_image(masks, polarity)
        cap = 400000
        det = cef.EfficientFeatures.create(cap, 1.2, 1, 0, 20, 0, 0)       # one level, NMS radius 0: every corner is kept
        kps, cnt = det.detectAsync(torch.from_numpy(img).cuda(), capacity=cap)
        torch.cuda.synchronize()
        n = int(cnt.item())
        st = det.lastLevelStats()[0]
        assert st["n_candidates"] == n < cap and n <= round(0.1 * img.size)   # neither capacity nor the 10 % cap cut anything
        loc = kps[0, :n].cpu().numpy().view(np.uint32)
        got = set(zip((loc & 0xffff).tolist(), (loc >> 16).tolist()))
        detected[masks] = [(int(x), int(y)) in got for x, y in zip(cx, cy)]
    t = table_from_predicate(lambda m: detected[m])
    assert hashlib.sha256(t.tobytes()).hexdigest() == PINS["c_table"]["sha256"]
    assert not detected[:504].any()


# ---- learned descriptor parameters: params/*.bin == the reference's header tables ------------------
PARAMS_DIR = os.path.join(os.path.dirname(HERE), "cuda-efficient-features_amd", "params")
PARAM_BLOBS = ["bad256", "bad512", "hashsift256", "hashsift512"]


@pytest.mark.parametrize("name", PARAM_BLOBS)
def test_param_blob_equals_reference_header_table(name):
    """params/<name>.bin hashes to the digest tools/pin_reference_tables.py took from the reference's header
    (bad.p256.h:27,94, bad.p512.h:209,340, hash_sift.p{256,512}.h:22), and to the md5 MANIFEST.json records."""
    blob = open(os.path.join(PARAMS_DIR, name + ".bin"), "rb").read()
    assert len(blob) == PINS[name]["bytes"]
    assert hashlib.sha256(blob).hexdigest() == PINS[name]["sha256"]
    manifest = json.load(open(os.path.join(PARAMS_DIR, "MANIFEST.json")))
    assert hashlib.md5(blob).hexdigest() == manifest[name + ".bin"]["blob_md5"]
    assert manifest[name + ".bin"]["bytes"] == len(blob)


def test_param_blobs_are_well_formed():
    """Structure the reference's code relies on: every BAD box lies inside the 32 x 32 patch (bad.cpp:151-155 never
    leaves it before the affine map), radii 1..7; the HashSIFT matrices are finite."""
    for n in (256, 512):
        blob = open(os.path.join(PARAMS_DIR, f"bad{n}.bin"), "rb").read()
        boxes = np.frombuffer(blob, dtype="<i4", count=5 * n).reshape(n, 5)
        thr = np.frombuffer(blob, dtype="<f4", offset=20 * n)
        assert thr.size == n and np.isfinite(thr).all()
        assert (boxes[:, :4] - boxes[:, 4:5] >= 0).all() and (boxes[:, :4] + boxes[:, 4:5] <= 31).all()
        assert boxes[:, 4].min() >= 1 and boxes[:, 4].max() <= 7
        w = np.fromfile(os.path.join(PARAMS_DIR, f"hashsift{n}.bin"), dtype="<f8")
        assert w.size == 129 * n and np.isfinite(w).all()


def test_library_embeds_the_pinned_blobs():
    """The blobs linked into libefx_hip.so (params_embed.S) are the pinned tables: read through the exported symbols,
    no device needed."""
    import ctypes
    import cef_loader
    lib = cef_loader.load().lib()
    for name in PARAM_BLOBS:
        n = PINS[name]["bytes"]
        arr = (ctypes.c_ubyte * n).in_dll(lib, "efx_blob_" + name)
        assert hashlib.sha256(bytes(arr)).hexdigest() == PINS[name]["sha256"], name


@pytest.mark.gpu
def test_loaded_describers_use_the_pinned_blobs():
    """On the GPU box: the library the describers run from embeds the pinned tables, and a BAD describer built from
    them reproduces the oracle (which reads params/*.bin) on a probe -- blob, library and oracle agree."""
    test_library_embeds_the_pinned_blobs()
    for name in PARAM_BLOBS:
        test_param_blob_equals_reference_header_table(name)
