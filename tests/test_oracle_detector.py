"""CPU checks of the oracle's detector restatement against independent numpy implementations of the same specs
 and against the numbers the survey extracted from the reference (pyramid table and
quotas: cuda_efficient_features.cpp:136-174).  No GPU."""
import math

import numpy as np
import pytest

from tools import synth

# circle of radius 3, k = 0 at (0, +3) walking towards +x (cuda_fast.cu:179-207)
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def brute_fast9(img, t, border):
    """FAST-9 by definition: >= 9 circularly contiguous ring pixels all > p+t or all < p-t (strict)."""
    h, w = img.shape
    I = img.astype(np.int32)
    out = []
    ys, xs = np.mgrid[border:h - border, border:w - border]
    p = I[border:h - border, border:w - border]
    ring = np.stack([I[border + dy:h - border + dy, border + dx:w - border + dx] for dx, dy in RING], axis=0)
    br = ring > p + t
    dk = ring < p - t
    res = np.zeros(p.shape, bool)
    for m in (br, dk):
        mm = np.concatenate([m, m[:8]], axis=0)
        for s in range(16):
            res |= mm[s:s + 9].all(axis=0)
    for y, x in zip(ys[res], xs[res]):
        out.append((x, y))
    return np.array(out, dtype=np.int16).reshape(-1, 2)


def test_pyramid_geometry_matches_survey_table(oracle):
    want = {(1080, 1920): [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521), (772, 434), (643, 362), (536, 301)],
            (2160, 3840): [(3840, 2160), (3200, 1800), (2667, 1500), (2222, 1250), (1852, 1042), (1543, 868), (1286, 723), (1072, 603)],
            (4320, 7680): [(7680, 4320), (6400, 3600), (5333, 3000), (4444, 2500), (3704, 2083), (3086, 1736), (2572, 1447), (2143, 1206)]}
    for (r, c), table in want.items():
        lr, lc, sc = oracle.pyramid_geometry(r, c)
        assert list(zip(lc, lr)) == table
    assert [float(s) for s in oracle.pyramid_geometry(100, 100)[2]][:3] == [1.0, float(np.float32(1.2)), float(np.float32(1.2) * np.float32(1.2))]


def test_level_quotas_match_survey(oracle):
    assert oracle.level_quotas(40000) == [8687, 7239, 6033, 5027, 4189, 3491, 2909, 2425]
    assert oracle.level_quotas(10000) == [2172, 1810, 1508, 1257, 1047, 873, 727, 606]
    assert sum(oracle.level_quotas(5000)) == 5000


@pytest.mark.parametrize("threshold", [5, 20, 60])
def test_fast9_equals_brute_force(oracle, threshold):
    img = synth.synth_frame(120, 160, seed=2, density=1.0)
    got = oracle.fast9_detect(img, threshold=threshold, border=15)
    want = brute_fast9(img, threshold, 15)
    assert got.shape == want.shape and np.array_equal(got, want)      # both in raster order
    noise = synth.noise_frame(64, 80, seed=1)
    assert np.array_equal(oracle.fast9_detect(noise, threshold=threshold, border=3), brute_fast9(noise, threshold, 3))


def test_integral_equals_cumsum(oracle):
    img = synth.noise_frame(57, 91, seed=5)
    got = oracle.integral(img)
    want = np.zeros((58, 92), np.int64)
    want[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
    assert np.array_equal(got.astype(np.int64), want)


def numpy_resize(src, drows, dcols, fused_weights=False):
    """Spec S5 in numpy: float32 weights, fused multiply-adds, same operation order.  fused_weights: the other reading of
    the four weights (EFX_S5_FUSED_WEIGHTS: fma(-o, f, i2) / fma(o, f, -i1) instead of subtracting the rounded o * f)."""
    srows, scols = src.shape
    fx = np.float32(1.0 / (dcols / scols))
    fy = np.float32(1.0 / (drows / srows))
    sx = np.arange(dcols, dtype=np.float32) * fx
    sy = np.arange(drows, dtype=np.float32) * fy
    x1 = np.minimum(np.floor(sx).astype(np.int64), scols - 1)
    y1 = np.minimum(np.floor(sy).astype(np.int64), srows - 1)
    x2, y2 = x1 + 1, y1 + 1
    x2r, y2r = np.minimum(x2, scols - 1), np.minimum(y2, srows - 1)
    S = src.astype(np.float32)
    if fused_weights:
        # o * f is exact in double (24-bit x 24-bit), and so is its difference with a small integer: one rounding, like fmaf
        ox, oy = np.arange(dcols, dtype=np.float64), np.arange(drows, dtype=np.float64)
        wx0 = (x2 - ox * np.float64(fx)).astype(np.float32)[None, :]
        wx1 = (ox * np.float64(fx) - x1).astype(np.float32)[None, :]
        wy0 = (y2 - oy * np.float64(fy)).astype(np.float32)[:, None]
        wy1 = (oy * np.float64(fy) - y1).astype(np.float32)[:, None]
    else:
        wx0 = (x2.astype(np.float32) - sx)[None, :]
        wx1 = (sx - x1.astype(np.float32))[None, :]
        wy0 = (y2.astype(np.float32) - sy)[:, None]
        wy1 = (sy - y1.astype(np.float32))[:, None]
    # fma(pixel, rounded weight product, out): the product of an 8-bit and a 24-bit number is exact in double, and so is
    # its sum with the accumulator unless a weight is below 2^-20 of the other terms (then the double rounding could matter
    # in principle; it does not on these inputs)
    fma = lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    out = np.zeros((drows, dcols), np.float32)
    out = fma(S[np.ix_(y1, x1)], wx0 * wy0, out)
    out = fma(S[np.ix_(y1, x2r)], wx1 * wy0, out)
    out = fma(S[np.ix_(y2r, x1)], wx0 * wy1, out)
    out = fma(S[np.ix_(y2r, x2r)], wx1 * wy1, out)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def test_resize_equals_numpy_spec(oracle):
    img = synth.synth_frame(203, 311, seed=8)
    for (dr, dc) in ((169, 259), (203, 311), (100, 100)):
        assert np.array_equal(oracle.resize_linear(img, dr, dc), numpy_resize(img, dr, dc))
    # the chain of pyramid_level is repeated resize_linear
    lr, lc, _ = oracle.pyramid_geometry(203, 311, nlevels=3)
    l1 = numpy_resize(img, lr[1], lc[1])
    l2 = numpy_resize(l1, lr[2], lc[2])
    assert np.array_equal(oracle.pyramid_level(img, 2), l2)


def test_s5_weight_switch(tmp_path):
    """whether nvcc also contracts the weight subtractions of opencv_contrib's resize_linear cannot be decided in
    this image, so the other reading is a build switch of the oracle and of the HIP library (-DEFX_S5_FUSED_WEIGHTS=1, one
    shared pair of helpers: efx_s5_w_hi / efx_s5_w_lo).  Both builds of the oracle equal the numpy restatement of their
    reading, the default build is the separate-subtraction one, and the two readings really differ (in weights by one ulp,
    in a few pixels of a level) -- so a maintainer with a CUDA box can tell them apart from one cv::cuda::resize dump."""
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "oracle", "efx_oracle.c")
    libs = {}
    for flag in (0, 1):
        so = str(tmp_path / f"oracle_s5_{flag}.so")
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-fPIC", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                               f"-DEFX_S5_FUSED_WEIGHTS={flag}", "-shared", "-o", so, src, "-lm"])
        libs[flag] = ctypes.CDLL(so)
        assert libs[flag].efxo_s5_fused_weights() == flag
    img = synth.powerlaw_frame(480, 640, seed=3)
    outs = {}
    for flag, lib in libs.items():
        dr, dc = 400, 533
        dst = np.zeros((dr, dc), np.uint8)
        lib.efxo_resize_linear(img.ctypes.data_as(ctypes.c_void_p), 480, 640, 640, dst.ctypes.data_as(ctypes.c_void_p), dr, dc, dc)
        assert np.array_equal(dst, numpy_resize(img, dr, dc, fused_weights=bool(flag))), f"EFX_S5_FUSED_WEIGHTS={flag}"
        outs[flag] = dst
    ndiff = int(np.count_nonzero(outs[0] != outs[1]))
    assert 0 < ndiff < outs[0].size // 100, ndiff          # distinguishable, and only in rounding ties


def test_gaussian_equals_numpy_spec(oracle):
    taps = oracle.gaussian_taps()
    e = np.exp(-(np.arange(7) - 3.0) ** 2 / 8.0)
    assert np.array_equal(taps, (e / e.sum()).astype(np.float32))
    img = synth.noise_frame(40, 50, seed=9)
    idx_c = np.abs(np.arange(-3, 50 + 3))
    idx_c = np.where(idx_c >= 50, 2 * 49 - idx_c, idx_c)          # reflect-101
    idx_r = np.abs(np.arange(-3, 40 + 3))
    idx_r = np.where(idx_r >= 40, 2 * 39 - idx_r, idx_r)
    # fma(a, b, c) on float32 values == float32(float64(a) * float64(b) + float64(c)): the product is exact in
    # double; the double rounding of the sum cannot hit a float32 tie for these operand widths in practice
    fma = lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    P = img.astype(np.float32)[:, idx_c]
    tmp = np.zeros((40, 50), np.float32)
    for j in range(7):
        tmp = fma(np.full((40, 50), taps[j], np.float32), P[:, j:j + 50], tmp)
    T = tmp[idx_r, :]
    out = np.zeros((40, 50), np.float32)
    for j in range(7):
        out = fma(np.full((40, 50), taps[j], np.float32), T[j:j + 40, :], out)
    want = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    assert np.array_equal(oracle.gaussian7(img), want)


def test_harris_equals_numpy_spec(oracle):
    img = synth.synth_frame(64, 64, seed=10, density=2.0)
    I = img.astype(np.int64)
    for (x, y) in [(20, 20), (31, 17), (40, 45), (16, 47)]:
        sxx = sxy = syy = 0
        for iy in range(-3, 4):
            for ix in range(-3, 4):
                w = I[y + iy - 1:y + iy + 2, x + ix - 1:x + ix + 2]
                dx = (w[0, 2] + 2 * w[1, 2] + w[2, 2]) - (w[0, 0] + 2 * w[1, 0] + w[2, 0])
                dy = (w[2, 0] + 2 * w[2, 1] + w[2, 2]) - (w[0, 0] + 2 * w[0, 1] + w[0, 2])
                sxx += dx * dx; sxy += dx * dy; syy += dy * dy
        f = np.float32
        scale = f(1.0) / f(4 * 7 * 255)
        K = scale * scale
        a, b, c = f(sxx) * K, f(syy) * K, f(sxy) * K
        want = (a * b - c * c) - f(0.04) * (a + b) * (a + b)
        assert oracle.harris(img, x, y) == want


def test_atan2_close_to_libm_and_quadrants(oracle):
    rng = np.random.default_rng(0)
    for m01, m10 in list(rng.integers(-2_000_000, 2_000_000, size=(500, 2))) + [(0, 0), (0, 5), (0, -5), (7, 0), (-7, 0), (3, 3), (-3, 3)]:
        got = float(oracle.atan2_deg(int(m01), int(m10)))
        want = math.degrees(math.atan2(m01, m10)) % 360.0 if (m01 or m10) else 0.0
        assert abs(got - want) < 1e-4 or abs(abs(got - want) - 360.0) < 1e-4


def test_nms_ties_and_quota_semantics(oracle):
    """Spec S3 on a hand-made frame: the full pipeline keeps a corner only if nothing at least as strong lies
    within the radius, and the quota keeps the strongest by (response, y, x)."""
    img = synth.synth_frame(300, 400, seed=12)
    full = oracle.detect_and_compute(img, nfeatures=100000, nlevels=1)
    k = oracle.unpack_keypoints(full["kps"])
    pts = np.stack([k["x"], k["y"]], 1).astype(np.int64)
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, 10 ** 9)
    assert (d2 >= 225).all(), "two survivors closer than the NMS radius"
    lim = oracle.detect_and_compute(img, nfeatures=20, nlevels=1)
    kl = oracle.unpack_keypoints(lim["kps"])
    order = np.lexsort((k["x"], k["y"], -k["response"]))[:20]
    want = set(zip(k["x"][order].tolist(), k["y"][order].tolist()))
    assert set(zip(kl["x"].tolist(), kl["y"].tolist())) == want
    # canonical order (S1) is preserved by the quota filter
    keyf = lambda x, y: ((y // 64) * 7 + (x // 64), ((y % 64) // 16) * 4 + (x % 64) // 16, y % 16, x % 16)
    keys = [keyf(int(x), int(y)) for x, y in zip(kl["x"], kl["y"])]
    assert keys == sorted(keys)


def test_ic_angle_matches_moments(oracle):
    img = synth.synth_frame(80, 80, seed=13, density=2.0)
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    x, y = 40, 41
    m01 = m10 = 0
    for dy in range(-15, 16):
        for dx in range(-umax[abs(dy)], umax[abs(dy)] + 1):
            v = int(img[y + dy, x + dx])
            m10 += dx * v
            m01 += dy * v
    assert oracle.ic_angle(img, x, y) == oracle.atan2_deg(m01, m10)


def test_empty_and_ragged_inputs(oracle):
    assert oracle.detect_and_compute(np.full((100, 120), 9, np.uint8), nfeatures=100)["n"] == 0
    r = oracle.detect_and_compute(synth.synth_frame(33, 47, seed=1), nfeatures=100, desc_type=oracle.BAD_256)
    assert r["n"] >= 0 and r["desc"].shape == (r["n"], 32)
    # keypoints completely outside the image, zero size, huge size: defined (clamped) behaviour, no crash
    img = synth.noise_frame(64, 64, seed=2)
    kps = np.array([[-50, -50, 31, 10], [200, 10, 31, 0], [32, 32, 0, 0], [32, 32, 500, 45], [0, 0, 31, -1]], np.float32)
    for nbits in (256, 512):
        assert oracle.bad_compute(img, kps, nbits).shape == (5, nbits // 8)
        assert oracle.hashsift_compute(img, kps, nbits).shape == (5, nbits // 8)


def test_thread_count_does_not_change_results(oracle):
    """The OpenMP loops of the oracle (multi-core CPU baseline of bench.py) write disjoint outputs."""
    img = synth.synth_frame(300, 400, seed=21)
    try:
        oracle.set_threads(1)
        a = oracle.detect_and_compute(img, nfeatures=2000, desc_type=oracle.HASH_SIFT_256)
        oracle.set_threads(4)
        b = oracle.detect_and_compute(img, nfeatures=2000, desc_type=oracle.HASH_SIFT_256)
        c = oracle.detect_and_compute(img, nfeatures=2000, desc_type=oracle.BAD_512)
        oracle.set_threads(1)
        d = oracle.detect_and_compute(img, nfeatures=2000, desc_type=oracle.BAD_512)
    finally:
        oracle.set_threads(1)
    assert np.array_equal(a["kps"].view(np.uint32), b["kps"].view(np.uint32)) and np.array_equal(a["desc"], b["desc"])
    assert np.array_equal(c["kps"].view(np.uint32), d["kps"].view(np.uint32)) and np.array_equal(c["desc"], d["desc"])
