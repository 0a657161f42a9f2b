"""A SECOND, independent restatement of the reference's CPU descriptors, in numpy, written from the reference text alone
(modules/efficient_features/src/bad.cpp, hash_sift.cpp) without consulting oracle/efx_oracle.c -- VERDICT r4 item 3: the C
oracle has a single transcriber, and nothing in this image can run the reference itself (no OpenCV).  Test infrastructure:
imported by tests/test_oracle_second_opinion.py only.

Every float expression is evaluated in IEEE binary32 in the reference's operation order (numpy float32 scalars / arrays round
after every operation; no contraction), integer conversions follow the C++ rules, and the libm calls of the reference
(cos / sin in double, cosf / sinf / expf / atan2f / sqrtf in float) go to the HOST's libm through ctypes -- numpy's own
vectorised transcendentals are not bit-identical to glibc's.  OpenCV primitives are restated from their documented
behaviour: cv::integral (int32 prefix sums with a leading zero row / column), cvFloor, cvRound (round half to even),
saturate_cast<uchar>, gemm on CV_32F (double accumulation).
"""
import ctypes
import ctypes.util
import os

import numpy as np

f32 = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("cosf", "sinf", "expf", "sqrtf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
for _n in ("cos", "sin"):
    getattr(_libm, _n).restype = ctypes.c_double
    getattr(_libm, _n).argtypes = [ctypes.c_double]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARAMS = os.path.join(ROOT, "cuda-efficient-features_amd", "params")


def bad_params(nbits):
    """params/bad{N}.bin: int32[N][5] {x1, x2, y1, y2, boxRadius} (the field order of BoxPairParams, bad.cpp:40-43), float32[N]
    thresholds; the blobs are digest-pinned on the reference headers (tests/test_reference_table_pins.py)."""
    raw = open(os.path.join(PARAMS, f"bad{nbits}.bin"), "rb").read()
    boxes = np.frombuffer(raw[:nbits * 20], dtype=np.int32).reshape(nbits, 5)
    thr = np.frombuffer(raw[nbits * 20:], dtype=np.float32)
    assert thr.shape == (nbits,)
    return boxes, thr


def hashsift_matrix(nbits):
    """params/hashsift{N}.bin: float64[N][129] -> float32, as Mat(N, 129, CV_64F).convertTo(CV_32F) (hash_sift.cpp:390-392)."""
    w = np.fromfile(os.path.join(PARAMS, f"hashsift{nbits}.bin"), dtype=np.float64).reshape(nbits, 129)
    return w.astype(np.float32)


def integral_image(img):
    """cv::integral(8U) -> CV_32S, (rows + 1) x (cols + 1), first row and column zero."""
    ii = np.zeros((img.shape[0] + 1, img.shape[1] + 1), np.int64)
    ii[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
    assert ii.max() < 2 ** 31
    return ii.astype(np.int32)


def _roundnum(x):
    """CV_ROUNDNUM(x) = (int)(x + 0.5f) (bad.cpp:27): float addition, truncation towards zero."""
    return np.trunc(x.astype(f32) + f32(0.5)).astype(np.int64)


def bad_describe(img, kps, nbits, scale_factor=1.0):
    """BAD_Impl::compute / computeBAD (bad.cpp:254-405) on {x, y, size, angle} keypoints; returns n x nbits / 8 bytes."""
    boxes, thr = bad_params(nbits)
    bx1, bx2, by1, by2, brad = (boxes[:, k].astype(f32) for k in range(5))
    ii = integral_image(img)
    ii64 = ii.astype(np.int64)
    iw, ih = ii.shape[1], ii.shape[0]                       # integralImg.cols / rows
    fw, fh = iw - 1, ih - 1                                 # frameSize (bad.cpp:331)
    sf = f32(scale_factor)
    out = np.zeros((len(kps), nbits // 8), np.uint8)
    for n, (x, y, size, angle) in enumerate(np.asarray(kps, dtype=np.float32)):
        # ---- rectifyBoxes (bad.cpp:114-157), patchSize 32 x 32 ----
        s = f32(sf * size) / f32(f32(0.5) * f32(32 + 32))
        if angle == f32(-1):
            m00 = s; m01 = f32(0)
            m02 = f32(f32(f32(-0.5) * s) * f32(32)) + x
            m10 = f32(0); m11 = s
            m12 = f32(f32(f32(-s) * f32(0.5)) * f32(32)) + y
        else:
            if angle >= 0:
                a = float(angle) * 0.017453292519943295      # float * double -> double
                cosine, sine = f32(_libm.cos(a)), f32(_libm.sin(a))
            else:
                cosine, sine = f32(1), f32(0)
            m00 = f32(s * cosine)
            m01 = f32(f32(-s) * sine)
            m02 = f32(f32(f32(f32(f32(-s) * cosine) + f32(s * sine)) * f32(32)) * f32(0.5)) + x
            m10 = f32(s * sine)
            m11 = f32(s * cosine)
            m12 = f32(f32(f32(f32(f32(-s) * sine) - f32(s * cosine)) * f32(32)) * f32(0.5)) + y
        m02, m12 = f32(m02), f32(m12)
        X1 = _roundnum((m00 * bx1 + m01 * by1).astype(f32) + m02)
        Y1 = _roundnum((m10 * bx1 + m11 * by1).astype(f32) + m12)
        X2 = _roundnum((m00 * bx2 + m01 * by2).astype(f32) + m02)
        Y2 = _roundnum((m10 * bx2 + m11 * by2).astype(f32) + m12)
        R = _roundnum(s * brad)
        # ---- isKeypointInTheBorder (bad.cpp:86-104): note the divisor (w + h), not its half ----
        sb = f32(sf * size) / f32(32 + 32)
        bw = f32(f32(f32(32) * sb) * f32(1.75)); bh = bw
        border = bool(x < bw or f32(x + bw) >= f32(fw) or y < bh or f32(y + bh) >= f32(fh))
        if border:
            # ---- computeBadResponse (bad.cpp:166-251): clamped boxes, float means ----
            def mean(cx, cy):
                x1 = cx - R
                x1 = np.where(x1 < 0, 0, np.where(x1 >= iw - 1, iw - 2, x1))
                y1 = cy - R
                y1 = np.where(y1 < 0, 0, np.where(y1 >= ih - 1, ih - 2, y1))
                x2 = cx + R + 1
                x2 = np.where(x2 <= 0, 1, np.where(x2 >= iw, iw - 1, x2))
                y2 = cy + R + 1
                y2 = np.where(y2 <= 0, 1, np.where(y2 >= ih, ih - 1, y2))
                ssum = (ii64[y1, x1] + ii64[y2, x2] - ii64[y1, x2] - ii64[y2, x1]).astype(np.int32).astype(f32)   # float(A + D - B - C)
                area = ((y2 - y1) * (x2 - x1)).astype(np.int32).astype(f32)
                return (ssum / area).astype(f32)
            resp = (mean(X1, Y1) - mean(X2, Y2)).astype(f32)
            bits = resp <= thr
        else:
            # ---- interior path (bad.cpp:358-401): integer box sums against threshold * side^2 ----
            side = 1 + (R << 1)

            def box(cx, cy):
                x1, y1, x2, y2 = cx - R, cy - R, cx + R + 1, cy + R + 1
                # (the reference indexes unchecked: a keypoint whose boxes leave the integral image -- possible for sizes below ~9,
                # where the border margin of 0.875 size is less than the boxes' reach plus rounding -- is undefined behaviour
                # there and refused here)
                if x1.min() < 0 or y1.min() < 0 or x2.max() >= iw or y2.max() >= ih:
                    raise ValueError("interior-path box outside the integral image (undefined in the reference)")
                return ii64[y1, x1] + ii64[y2, x2] - ii64[y1, x2] - ii64[y2, x1]
            area = (box(X1, Y1) - box(X2, Y2)).astype(np.int32)
            bits = area.astype(f32) <= (thr * (side * side).astype(np.int32).astype(f32)).astype(f32)     # int <= float: the int converts
        out[n] = np.packbits(bits.astype(np.uint8))          # bit_idx = 7 - boxIdx % 8: MSB first
    return out


# ---------------------------------------------------------------------------------------------------------------------
# HashSIFT (hash_sift.cpp)
# ---------------------------------------------------------------------------------------------------------------------
def _cvfloor(a):
    return np.floor(a).astype(np.int64)


def hashsift_patch(img, kp, crop_scale=1.0):
    """rectifyPatch + warpAffineLinear (hash_sift.cpp:68-138), 32 x 32."""
    x0, y0, size, angle = (f32(v) for v in kp)
    rows, cols = img.shape
    s = f32(f32(crop_scale) * size) / f32(f32(0.5) * f32(32 + 32))
    theta = f32(f32(f32(np.pi) * angle) / f32(180))
    cost = f32(s * (f32(_libm.cosf(theta)) if angle >= 0 else f32(1)))
    sint = f32(s * (f32(_libm.sinf(theta)) if angle >= 0 else f32(0)))
    M00, M01 = cost, f32(-sint)
    M02 = f32(f32(f32(f32(-cost) + sint) * f32(32)) / f32(2)) + x0
    M10, M11 = sint, cost
    M12 = f32(f32(f32(f32(-sint) - cost) * f32(32)) / f32(2)) + y0
    M02, M12 = f32(M02), f32(M12)
    xs = np.arange(32, dtype=np.float32)[None, :]
    ys = np.arange(32, dtype=np.float32)[:, None]
    u = ((M00 * xs).astype(f32) + (M01 * ys).astype(f32)).astype(f32) + M02
    v = ((M10 * xs).astype(f32) + (M11 * ys).astype(f32)).astype(f32) + M12
    u, v = u.astype(f32), v.astype(f32)
    ui, vi = _cvfloor(u), _cvfloor(v)
    ok = (ui >= 0) & (ui + 1 < cols) & (vi >= 0) & (vi + 1 < rows)
    uc, vc = np.where(ok, ui, 0), np.where(ok, vi, 0)
    p00 = img[vc, uc].astype(f32); p01 = img[vc, np.minimum(uc + 1, cols - 1)].astype(f32)
    p10 = img[np.minimum(vc + 1, rows - 1), uc].astype(f32); p11 = img[np.minimum(vc + 1, rows - 1), np.minimum(uc + 1, cols - 1)].astype(f32)
    du = (u - ui.astype(f32)).astype(f32); dv = (v - vi.astype(f32)).astype(f32)
    one = f32(1)
    tmp0 = (((one - du).astype(f32) * p00).astype(f32) + (du * p01).astype(f32)).astype(f32)
    tmp1 = (((one - du).astype(f32) * p10).astype(f32) + (du * p11).astype(f32)).astype(f32)
    tmp2 = (((one - dv).astype(f32) * tmp0).astype(f32) + (dv * tmp1).astype(f32)).astype(f32)
    val = np.minimum(np.trunc((tmp2 + f32(0.5)).astype(f32)).astype(np.int64), 255)
    return np.where(ok, val, 0).astype(np.uint8)


_tables = {}


def _pixel_tables():
    """What computePatchSIFT (hash_sift.cpp:200-331) derives from the patch geometry alone, h = w = 32, kpScale = 1.f / 6."""
    if _tables:
        return _tables
    kp_scale = f32(1) / f32(6)
    h = w = 32
    kp_radius = f32(f32(kp_scale * f32(h)) * f32(0.5))
    kernel_sigma = f32(f32(f32(f32(0.5) * f32(4)) * f32(3)) * kp_radius)
    dist_scale = f32(-1) / f32(f32(f32(2) * kernel_sigma) * kernel_sigma)
    cx = f32(f32(0.5) * f32(30)); cy = cx
    cellh = f32(f32(3) * f32(f32(kp_scale * f32(h)) * f32(0.5)))
    scale_r = f32(1) / cellh; scale_c = scale_r
    scale_o = f32(8) / f32(2 * np.pi)
    half = f32(f32(0.5) * f32(32))
    bin0 = f32(4 // 2 - f32(0.5))
    idx = np.arange(30)
    # getRBin(y + 1) == getCBin(x + 1) (h == w); r - halfh with r an int: the int converts to float exactly
    rb = ((scale_r * ((idx + 1).astype(f32) - half).astype(f32)).astype(f32) + bin0).astype(f32)
    ri = _cvfloor(rb); rf = (rb - ri.astype(f32)).astype(f32)
    mag_scale = np.zeros((30, 30), np.float32)
    for y in range(30):
        for x in range(30):
            dxs = f32(f32(x) - cx); dys = f32(f32(y) - cy)
            nsq = f32(f32(dxs * dxs) + f32(dys * dys))
            mag_scale[y, x] = _libm.expf(f32(dist_scale * nsq))
    _tables.update(ri=ri, rf=rf, mag_scale=mag_scale, scale_o=scale_o)
    return _tables


def _normalize(d):
    """normalize (hash_sift.cpp:150-160) on the rows of d: the sum of squares is accumulated serially, in float."""
    ssum = np.zeros(d.shape[0], np.float32)
    for i in range(d.shape[1]):
        ssum = (ssum + (d[:, i] * d[:, i]).astype(f32)).astype(f32)
    norm = np.maximum(np.sqrt(ssum).astype(f32), np.finfo(np.float32).eps).astype(f32)
    scale = (f32(1) / norm).astype(f32)
    return (d * scale[:, None]).astype(f32)


def hashsift_vectors(img, kps, crop_scale=1.0):
    """computePatchSIFTs (hash_sift.cpp:333-351): n x 129 float responses (element 0 is the constant 1)."""
    T = _pixel_tables()
    kps = np.asarray(kps, dtype=np.float32).reshape(-1, 4)
    n = len(kps)
    patches = np.stack([hashsift_patch(img, k, crop_scale) for k in kps]).astype(np.int32)     # n x 32 x 32
    hist = np.zeros((n, 6, 6, 10), np.float32)
    rows = np.arange(n)
    atan2f = np.vectorize(lambda a, b: _libm.atan2f(a, b), otypes=[np.float32])
    for y in range(30):
        ri, rf = int(T["ri"][y]), T["rf"][y]
        for x in range(30):
            dx = (patches[:, y + 1, x + 2] - patches[:, y + 1, x]).astype(f32)
            dy = (patches[:, y, x + 1] - patches[:, y + 2, x + 1]).astype(f32)
            mag = (T["mag_scale"][y, x] * np.sqrt(((dx * dx).astype(f32) + (dy * dy).astype(f32)).astype(f32)).astype(f32)).astype(f32)
            ori = atan2f(dy, dx)
            ci, cf = int(T["ri"][x]), T["rf"][x]
            ob = (T["scale_o"] * ori).astype(f32)
            oi = _cvfloor(ob)
            of = (ob - oi.astype(f32)).astype(f32)
            oi = np.where(oi < 0, oi + 8, oi)
            oi = np.where(oi >= 8, oi - 8, oi)

            def distribute(value, weight):
                v1 = (weight * value).astype(f32)
                return (value - v1).astype(f32), v1
            v0, v1 = distribute(mag, rf)
            v00, v01 = distribute(v0, cf)
            v10, v11 = distribute(v1, cf)
            v000, v001 = distribute(v00, of); v010, v011 = distribute(v01, of)
            v100, v101 = distribute(v10, of); v110, v111 = distribute(v11, of)
            for (r_, c_, o_, val) in ((ri + 1, ci + 1, oi, v000), (ri + 1, ci + 1, oi + 1, v001), (ri + 1, ci + 2, oi, v010), (ri + 1, ci + 2, oi + 1, v011),
                                      (ri + 2, ci + 1, oi, v100), (ri + 2, ci + 1, oi + 1, v101), (ri + 2, ci + 2, oi, v110), (ri + 2, ci + 2, oi + 1, v111)):
                hist[rows, r_, c_, o_] = (hist[rows, r_, c_, o_] + val).astype(f32)
    d = np.zeros((n, 128), np.float32)
    for r in range(4):
        for c in range(4):
            cell = hist[:, r + 1, c + 1, :].copy()
            cell[:, 0] = (cell[:, 0] + cell[:, 8]).astype(f32)
            cell[:, 1] = (cell[:, 1] + cell[:, 9]).astype(f32)
            d[:, (r * 4 + c) * 8:(r * 4 + c) * 8 + 8] = cell[:, :8]
    d = _normalize(d)
    d = np.minimum(d, f32(0.2)).astype(f32)
    d = _normalize(d)
    # saturate_cast<uchar>(float) = cvRound (round half to even) then clamp
    q = np.clip(np.rint((f32(512) * d).astype(f32)), 0, 255).astype(f32)
    return np.concatenate([np.ones((n, 1), np.float32), q], axis=1), (f32(512) * d).astype(f32)


def hashsift_bits(resp, nbits):
    """matmulAndSign (hash_sift.cpp:353-378): cv::gemm on CV_32F accumulates in double; bit = product > 0, MSB first.
    Returns (bytes, T as float64)."""
    W = hashsift_matrix(nbits)
    T = resp.astype(np.float64) @ W.astype(np.float64).T
    return np.packbits((T.astype(np.float32) > 0).astype(np.uint8), axis=1), T
