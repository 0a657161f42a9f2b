"""ctypes binding of the CPU ORACLE (oracle/libefx_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (cuda-efficient-features_amd/) never does.  Parity status: "parity unpinned"
(see oracle/efx_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libefx_oracle.so")
PARAMS_DIR = os.path.join(_HERE, "..", "cuda-efficient-features_amd", "params")

BAD_256, BAD_512, HASH_SIFT_256, HASH_SIFT_512 = 0, 1, 2, 3
MAX_LEVELS = 32


def build(force=False):
    src = os.path.join(_HERE, "efx_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "efx_oracle.h"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("first_level", C.c_int), ("fast_threshold", C.c_int), ("nonmax_radius", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("n_candidates", C.c_int * MAX_LEVELS), ("n_after_cap", C.c_int * MAX_LEVELS),
                ("n_after_nms", C.c_int * MAX_LEVELS), ("n_kept", C.c_int * MAX_LEVELS)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.efxo_harris.restype = C.c_float
        _lib.efxo_ic_angle.restype = C.c_float
        _lib.efxo_atan2_deg.restype = C.c_float
    return _lib


def _u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def load_bad_params(nbits):
    path = os.path.join(PARAMS_DIR, f"bad{nbits}.bin")
    boxes = np.fromfile(path, dtype=np.int32, count=nbits * 5).reshape(nbits, 5).copy()
    thr = np.fromfile(path, dtype=np.float32, offset=nbits * 5 * 4).copy()
    assert thr.shape == (nbits,)
    return boxes, thr


def load_hashsift_weights(nbits):
    path = os.path.join(PARAMS_DIR, f"hashsift{nbits}.bin")
    w = np.fromfile(path, dtype=np.float64).reshape(nbits, 129)
    return np.ascontiguousarray(w.astype(np.float32))   # convertTo(CV_32F), hash_sift.cpp:390-392


def integral(img):
    img = _u8(img)
    out = np.empty((img.shape[0] + 1, img.shape[1] + 1), dtype=np.int32)
    lib().efxo_integral(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(out))
    return out


def bad_compute(img, kps, nbits, scale_factor=1.0):
    """kps: (n,4) float32 {x, y, size, angle}."""
    img = _u8(img)
    kps = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 4)
    boxes, thr = load_bad_params(nbits)
    desc = np.zeros((kps.shape[0], nbits // 8), dtype=np.uint8)
    lib().efxo_bad_compute(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(kps), kps.shape[0],
                           C.c_float(scale_factor), _p(boxes), _p(thr), nbits, _p(desc))
    return desc


def hashsift_patch(img, kp4, crop_scale=1.0):
    img = _u8(img)
    kp4 = np.ascontiguousarray(kp4, dtype=np.float32).reshape(4)
    patch = np.zeros((32, 32), dtype=np.uint8)
    lib().efxo_hashsift_patch(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(kp4),
                              C.c_float(crop_scale), _p(patch))
    return patch


def hashsift_responses(img, kps, crop_scale=1.0):
    img = _u8(img)
    kps = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 4)
    resp = np.zeros((kps.shape[0], 129), dtype=np.float32)
    lib().efxo_hashsift_responses(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(kps), kps.shape[0],
                                  C.c_float(crop_scale), _p(resp))
    return resp


def hashsift_responses_fixedpoint(img, kps, crop_scale=1.0):
    img = _u8(img)
    kps = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 4)
    resp = np.zeros((kps.shape[0], 129), dtype=np.float32)
    lib().efxo_hashsift_responses_fixedpoint(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(kps), kps.shape[0],
                                  C.c_float(crop_scale), _p(resp))
    return resp


def hashsift_responses_model(img, kps, mode, crop_scale=1.0):
    """mode: bit 0 = fixed-point histogram, bit 1 = tree-ordered norms (0 = reference arithmetic, 3 = device model)."""
    img = _u8(img)
    kps = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 4)
    resp = np.zeros((kps.shape[0], 129), dtype=np.float32)
    lib().efxo_hashsift_responses_model(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(kps), kps.shape[0],
                                        C.c_float(crop_scale), int(mode), _p(resp))
    return resp


def hashsift_project(resp, nbits):
    resp = np.ascontiguousarray(resp, dtype=np.float32).reshape(-1, 129)
    W = load_hashsift_weights(nbits)
    T = np.zeros((resp.shape[0], nbits), dtype=np.float32)
    desc = np.zeros((resp.shape[0], nbits // 8), dtype=np.uint8)
    lib().efxo_hashsift_project(_p(resp), resp.shape[0], _p(W), nbits, _p(T), _p(desc))
    return T, desc


def hashsift_compute(img, kps, nbits, crop_scale=1.0):
    resp = hashsift_responses(img, kps, crop_scale)
    return hashsift_project(resp, nbits)[1]


def set_threads(n):
    """OpenMP threads of the oracle's loops (results do not depend on it); returns the previous value."""
    prev = lib().efxo_get_threads()
    lib().efxo_set_threads(int(n))
    return prev


def has_arc9(mask16):
    return int(lib().efxo_has_arc9(C.c_uint(int(mask16))))


def ic_umax():
    u = (C.c_int * 17)()
    lib().efxo_ic_umax(u)
    return list(u)


def calc_umax(patch_size):
    u = (C.c_int * (patch_size // 2 + 2))()
    lib().efxo_calc_umax(int(patch_size), u)
    return list(u)


def fast_atan2(y, x):
    lib().efxo_fast_atan2.restype = C.c_float
    lib().efxo_fast_atan2.argtypes = [C.c_float, C.c_float]
    return np.float32(lib().efxo_fast_atan2(float(y), float(x)))


def ic_angles(img, kp4, patch_size):
    """ICAngles of hpatches_description.cpp:128-162; returns a copy of kp4 with the angle column filled."""
    img = _u8(img)
    k = np.ascontiguousarray(kp4, dtype=np.float32).copy()
    lib().efxo_ic_angles(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(k), len(k), int(patch_size))
    return k


def bgr2gray(img):
    """H x W x 3|4 uint8 -> H x W uint8 (spec S11)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros(img.shape[:2], dtype=np.uint8)
    lib().efxo_bgr2gray(_p(img), img.shape[0], img.shape[1], img.strides[0], img.shape[2], _p(out), out.strides[0])
    return out


def pyramid_geometry(rows, cols, scale_factor=1.2, nlevels=8):
    lr = (C.c_int * nlevels)()
    lc = (C.c_int * nlevels)()
    sc = (C.c_float * nlevels)()
    lib().efxo_pyramid_geometry(rows, cols, C.c_float(scale_factor), nlevels, lr, lc, sc)
    return list(lr), list(lc), [np.float32(v) for v in sc]


def level_quotas(total, scale_factor=1.2, nlevels=8):
    q = (C.c_int * nlevels)()
    lib().efxo_level_quotas(total, C.c_float(scale_factor), nlevels, q)
    return list(q)


def resize_linear(src, drows, dcols):
    src = _u8(src)
    dst = np.zeros((drows, dcols), dtype=np.uint8)
    lib().efxo_resize_linear(_p(src), src.shape[0], src.shape[1], src.strides[0], _p(dst), drows, dcols, dcols)
    return dst


def pyramid_level(img, level, scale_factor=1.2):
    img = _u8(img)
    lr, lc, _ = pyramid_geometry(img.shape[0], img.shape[1], scale_factor, level + 1)
    dst = np.zeros((lr[level], lc[level]), dtype=np.uint8)
    rc = lib().efxo_pyramid_level(_p(img), img.shape[0], img.shape[1], img.strides[0], C.c_float(scale_factor),
                                  level, _p(dst))
    assert rc == 0
    return dst


def gaussian_taps():
    t = (C.c_float * 7)()
    lib().efxo_gaussian_taps(t)
    return np.array(list(t), dtype=np.float32)


def gaussian7(src):
    src = _u8(src)
    dst = np.zeros_like(src)
    lib().efxo_gaussian7(_p(src), src.shape[0], src.shape[1], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast9_detect(img, threshold=20, border=15):
    img = _u8(img)
    cap = img.size
    xy = np.zeros((cap, 2), dtype=np.int16)
    n = lib().efxo_fast9_detect(_p(img), img.shape[0], img.shape[1], img.strides[0], threshold, border, _p(xy), cap)
    return xy[:n].copy()


def harris(img, x, y):
    img = _u8(img)
    return np.float32(lib().efxo_harris(_p(img), img.strides[0], int(x), int(y)))


def ic_angle(img, x, y):
    img = _u8(img)
    return np.float32(lib().efxo_ic_angle(_p(img), img.strides[0], int(x), int(y)))


def atan2_deg(m01, m10):
    return np.float32(lib().efxo_atan2_deg(int(m01), int(m10)))


def detect_and_compute(img, nfeatures=5000, scale_factor=1.2, nlevels=8, first_level=0, fast_threshold=20,
                       nonmax_radius=15, desc_type=-1, capacity=None, mask=None):
    """Returns dict(kps=(5,N) float32 raw rows, desc=(N,nbytes) or None, lvl_xy=(2,N) int16, stats)."""
    img = _u8(img)
    if capacity is None:
        capacity = max(nfeatures, 1)
    p = Params(nfeatures, scale_factor, nlevels, first_level, fast_threshold, nonmax_radius)
    kps = np.zeros((5, capacity), dtype=np.float32)
    lvl = np.zeros((2, capacity), dtype=np.int16)
    st = Stats()
    a = b = None
    desc = None
    if desc_type in (BAD_256, BAD_512):
        nbits = 256 if desc_type == BAD_256 else 512
        a, b = load_bad_params(nbits)
        desc = np.zeros((capacity, nbits // 8), dtype=np.uint8)
    elif desc_type in (HASH_SIFT_256, HASH_SIFT_512):
        nbits = 256 if desc_type == HASH_SIFT_256 else 512
        a = load_hashsift_weights(nbits)
        desc = np.zeros((capacity, nbits // 8), dtype=np.uint8)
    if mask is not None:
        mask = _u8(mask)
        assert mask.shape == img.shape
    n = lib().efxo_detect_and_compute_masked(_p(img), img.shape[0], img.shape[1], img.strides[0],
                                             _p(mask) if mask is not None else None, mask.strides[0] if mask is not None else 0,
                                             C.byref(p), desc_type,
                                             _p(a) if a is not None else None, _p(b) if b is not None else None,
                                             _p(kps), _p(desc) if desc is not None else None, _p(lvl), capacity,
                                             C.byref(st))
    if n < 0:
        raise ValueError("efxo_detect_and_compute: bad arguments")
    stats = {k: list(getattr(st, k))[:nlevels] for k in ("n_candidates", "n_after_cap", "n_after_nms", "n_kept")}
    return dict(n=n, kps=kps[:, :n].copy(), desc=None if desc is None else desc[:n].copy(),
                lvl_xy=lvl[:, :n].copy(), stats=stats)


def compute_provided(img, kps, desc_type, scale_factor=1.2, nlevels=8):
    """detectAndCompute(useProvidedKeypoints=True), spec S13: kps is the (5, N) raw float32 matrix."""
    img = _u8(img)
    kps = np.ascontiguousarray(kps, dtype=np.float32)
    n = kps.shape[1]
    nbits = 256 if desc_type in (BAD_256, HASH_SIFT_256) else 512
    if desc_type in (BAD_256, BAD_512):
        a, b = load_bad_params(nbits)
    else:
        a, b = load_hashsift_weights(nbits), None
    desc = np.zeros((max(n, 1), nbits // 8), dtype=np.uint8)
    p = Params(0, scale_factor, nlevels, 0, 20, 15)
    rc = lib().efxo_compute_provided(_p(img), img.shape[0], img.shape[1], img.strides[0], C.byref(p), desc_type,
                                     _p(a), _p(b) if b is not None else None, _p(kps), kps.shape[1], n, _p(desc))
    if rc:
        raise ValueError("efxo_compute_provided: bad arguments")
    return desc[:n]


def unpack_keypoints(kps):
    """(5,N) raw rows -> dict of x, y (int16), response, angle, octave (int32), size."""
    loc = kps[0].view(np.uint32)
    x = (loc & 0xFFFF).astype(np.uint16).view(np.int16)
    y = (loc >> 16).astype(np.uint16).view(np.int16)
    return dict(x=x, y=y, response=kps[1].copy(), angle=kps[2].copy(), octave=kps[3].view(np.int32).copy(),
                size=kps[4].copy())
