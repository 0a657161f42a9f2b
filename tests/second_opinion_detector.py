"""A SECOND restatement of the whole detectAndCompute flow, in numpy, assembled from the reference text
(modules/cuda_efficient_features/src/cuda_efficient_features.cpp:136-174, 225-321; cuda_efficient_features.cu:54-172, 236-248;
cuda_fast.cu:168-222) and from DESIGN.md section 3's spec decisions S1-S8 -- NOT from oracle/efx_oracle.c.  The pieces that were
already independent numpy checks of single stages (resize, FAST by definition, Gaussian, Harris: tests/test_oracle_detector.py) are
chained here with the parts that had no second opinion yet: level geometry and quotas, the candidate cap (S2), the radius
suppression as the literal O(n^2) predicate of IsMaxPoint, the quota's (response, y, x) order (S3), the canonical output order (S1),
the intensity-centroid moments, scalePoints, the 5 x N row encoding, and the describer on the BLURRED level at level coordinates.
Test infrastructure: imported by tests/test_oracle_second_opinion_detector.py only.  Brute force everywhere: small frames.
"""
import math

import numpy as np

from tests import second_opinion as so

f32 = np.float32
PATCH_SIZE, HALF_PATCH_SIZE = 31, 15                      # cuda_efficient_features.cpp:33-34
CORNER_DENSITY = 0.1                                      # :35
CELL, TILE = 16, 64                                       # CELL_SIZE of the suppression grid (.cu); the canonical order's tile (spec S1)
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]               # cuda_fast.cu:179-207
U_MAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3, 0]      # cuda_efficient_features.cu:143


def cv_round(x):
    """cvRound: round half to even (lrint / cvtsd2si)."""
    return int(np.rint(x))


def pyramid_geometry(rows, cols, scale_factor, nlevels):
    """calcImagePyramid, .cpp:136-157: `scale *= scaleFactor` in float, sizes = cvRound(invScale * rows / cols) (float products)."""
    scale = f32(1.0)
    out = [(rows, cols, scale)]
    for _ in range(1, nlevels):
        scale = f32(scale * f32(scale_factor))
        inv = f32(f32(1.0) / scale)
        out.append((cv_round(f32(inv * f32(rows))), cv_round(f32(inv * f32(cols))), scale))
    return out


def level_quotas(total, scale_factor, nlevels):
    """calcNumFeaturesPerLevel, .cpp:159-174: double arithmetic on the FLOAT scale factor."""
    factor = 1.0 / float(f32(scale_factor))
    n = total * (1 - factor) / (1 - factor ** nlevels)
    out, s = [], 0
    for _ in range(nlevels - 1):
        out.append(cv_round(n)); s += out[-1]
        n *= factor
    out.append(max(total - s, 0))
    return out


def _fma(a, b, c):
    # fmaf on float32 operands: the product is exact in double; the double rounding of the sum does not hit a float32 tie for
    # these operand widths (8-bit pixel x 24-bit weight + 24-bit accumulator)
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def resize_linear(src, drows, dcols):
    """Spec S5 (cv::cuda::resize INTER_LINEAR as this build reads it): src = dst * f, f = (float)(1 / (dsize / ssize)) with the
    division in double, floor, the +1 neighbour clamped, four ROUNDED weight products, fma(pixel, w, acc) in (00, 01, 10, 11)
    order, round half even, saturate."""
    srows, scols = src.shape
    fx = f32(1.0 / (dcols / scols)); fy = f32(1.0 / (drows / srows))
    sx = (np.arange(dcols, dtype=np.float32) * fx).astype(np.float32)
    sy = (np.arange(drows, dtype=np.float32) * fy).astype(np.float32)
    x1 = np.minimum(np.floor(sx).astype(np.int64), scols - 1); y1 = np.minimum(np.floor(sy).astype(np.int64), srows - 1)
    x2, y2 = x1 + 1, y1 + 1
    x2c, y2c = np.minimum(x2, scols - 1), np.minimum(y2, srows - 1)
    wx0 = (x2.astype(np.float32) - sx)[None, :]; wx1 = (sx - x1.astype(np.float32))[None, :]
    wy0 = (y2.astype(np.float32) - sy)[:, None]; wy1 = (sy - y1.astype(np.float32))[:, None]
    S = src.astype(np.float32)
    acc = np.zeros((drows, dcols), np.float32)
    acc = _fma(S[np.ix_(y1, x1)], (wx0 * wy0).astype(np.float32), acc)
    acc = _fma(S[np.ix_(y1, x2c)], (wx1 * wy0).astype(np.float32), acc)
    acc = _fma(S[np.ix_(y2c, x1)], (wx0 * wy1).astype(np.float32), acc)
    acc = _fma(S[np.ix_(y2c, x2c)], (wx1 * wy1).astype(np.float32), acc)
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def gaussian7(img):
    """Spec S6: createGaussianFilter(7 x 7, sigma 2, BORDER_REFLECT_101): taps exp(-(i - 3)^2 / 8) normalised in double -> float;
    row pass u8 -> float, column pass float -> u8 (round half even); acc = fma(tap_j, v_j, acc), j = 0 .. 6."""
    rows, cols = img.shape
    e = np.exp(-(np.arange(7) - 3.0) ** 2 / 8.0)
    taps = (e / e.sum()).astype(np.float32)
    refl = lambda n: (lambda i: np.where(i >= n, 2 * (n - 1) - i, i))(np.abs(np.arange(-3, n + 3)))
    P = img.astype(np.float32)[:, refl(cols)]
    tmp = np.zeros((rows, cols), np.float32)
    for j in range(7):
        tmp = _fma(taps[j], P[:, j:j + cols], tmp)
    T = tmp[refl(rows), :]
    out = np.zeros((rows, cols), np.float32)
    for j in range(7):
        out = _fma(taps[j], T[j:j + rows, :], out)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def fast9(img, t, border=HALF_PATCH_SIZE):
    """FAST-9 by definition (cuda_fast.cu:168-222 with its c_table = ">= 9 circularly contiguous"; strict compares, diffType :36-40)
    inside the border mask (createMask, .cpp:176-182).  Returns the corner bitmap."""
    h, w = img.shape
    res = np.zeros((h, w), bool)
    if h <= 2 * border or w <= 2 * border:
        return res
    I = img.astype(np.int32)
    p = I[border:h - border, border:w - border]
    ring = np.stack([I[border + dy:h - border + dy, border + dx:w - border + dx] for dx, dy in RING], axis=0)
    hit = np.zeros(p.shape, bool)
    for m in (ring > p + t, ring < p - t):
        mm = np.concatenate([m, m[:8]], axis=0)
        for s in range(16):
            hit |= mm[s:s + 9].all(axis=0)
    res[border:h - border, border:w - border] = hit
    return res


def canonical_order(x, y, tiles_x):
    """Spec S1: 64 x 64 tiles row-major -> 16 x 16 cells row-major inside the tile -> raster inside the cell."""
    x = np.asarray(x, np.int64); y = np.asarray(y, np.int64)
    key = (((y // TILE) * tiles_x + x // TILE) << 12) | ((((y % TILE) // CELL) * 4 + (x % TILE) // CELL) << 8) | ((y % CELL) << 4) | (x % CELL)
    return np.argsort(key, kind="stable")


def harris(img, xs, ys):
    """calcResponse (.cu:99-139) under spec S4: exact integer sums of the unscaled Sobel products over the 7 x 7 block, then
    a = f(Sxx) K, b = f(Syy) K, c = f(Sxy) K (K = SCALE^2), R = (a b - c c) - (0.04 (a + b)) (a + b) in float, no contraction."""
    I = img.astype(np.int64)
    h, w = I.shape
    dx = np.zeros((h, w), np.int64); dy = np.zeros((h, w), np.int64)
    dx[1:-1, 1:-1] = (I[:-2, 2:] + 2 * I[1:-1, 2:] + I[2:, 2:]) - (I[:-2, :-2] + 2 * I[1:-1, :-2] + I[2:, :-2])
    dy[1:-1, 1:-1] = (I[2:, :-2] + 2 * I[2:, 1:-1] + I[2:, 2:]) - (I[:-2, :-2] + 2 * I[:-2, 1:-1] + I[:-2, 2:])

    def box7(a):
        c = np.zeros((h + 1, w + 1), np.int64); c[1:, 1:] = a.cumsum(0).cumsum(1)
        return lambda x, y: c[y + 4, x + 4] - c[y - 3, x + 4] - c[y + 4, x - 3] + c[y - 3, x - 3]
    xs = np.asarray(xs, np.int64); ys = np.asarray(ys, np.int64)
    sxx, syy, sxy = box7(dx * dx)(xs, ys), box7(dy * dy)(xs, ys), box7(dx * dy)(xs, ys)
    scale = f32(1.0) / f32(4 * 7 * 255)
    K = f32(scale * scale)
    a = (sxx.astype(np.float32) * K).astype(np.float32); b = (syy.astype(np.float32) * K).astype(np.float32)
    c = (sxy.astype(np.float32) * K).astype(np.float32)
    det = ((a * b).astype(np.float32) - (c * c).astype(np.float32)).astype(np.float32)
    tr = (a + b).astype(np.float32)
    return (det - ((f32(0.04) * tr).astype(np.float32) * tr).astype(np.float32)).astype(np.float32)


def radius_suppression(x, y, resp, radius):
    """radiusSuppressionKernel + IsMaxPoint (.cu:62-97): point i survives iff NO other point j has response_i <= response_j and
    dx^2 + dy^2 < cvCeil(r * r).  (The reference looks at the cells within ceil(r / 16) of i's cell: every point closer than r
    is in one of them.)  The literal O(n^2) predicate."""
    n = len(x)
    keep = np.ones(n, bool)
    X = np.asarray(x, np.int64); Y = np.asarray(y, np.int64); R = np.asarray(resp, np.float32)
    r2 = int(math.ceil(radius * radius))
    for i0 in range(0, n, 512):
        sl = slice(i0, min(i0 + 512, n))
        d2 = (X[sl, None] - X[None, :]) ** 2 + (Y[sl, None] - Y[None, :]) ** 2
        rival = (R[sl, None] <= R[None, :]) & (d2 < r2)
        rival[np.arange(sl.stop - sl.start), np.arange(sl.start, sl.stop)] = False          # idx1 == idx2: continue
        keep[sl] = ~rival.any(axis=1)
    return keep


def ic_angle(img, x, y):
    """IC_Angle + convertToDegree (.cu:54-60, 141-172): integer moments over the radius-15 circular patch; the angle itself under
    spec S7 (double atan2 -> [0, 2 pi) -> degrees -> float; the reference's device atan2f is not reproducible off the device)."""
    I = img.astype(np.int64)
    m01 = m10 = 0
    for d in range(-HALF_PATCH_SIZE, HALF_PATCH_SIZE + 1):
        m10 += d * int(I[y, x + d])
    for dy in range(1, HALF_PATCH_SIZE + 1):
        ysum = 0
        for d in range(-U_MAX[dy], U_MAX[dy] + 1):
            t, b = int(I[y - dy, x + d]), int(I[y + dy, x + d])
            ysum += b - t
            m10 += d * (b + t)
        m01 += dy * ysum
    if m01 == 0 and m10 == 0:
        return f32(0.0), (m01, m10)
    a = math.atan2(float(m01), float(m10))
    if a < 0:
        a += 2 * math.pi
    return f32(a * (180.0 / math.pi)), (m01, m10)


def detect_and_compute(img, nfeatures, scale_factor=1.2, nlevels=8, fast_threshold=20, nonmax_radius=15, bad_bits=0, hashsift=False):
    """detectAndComputeAsync (.cpp:225-321), firstLevel 0, no mask.  Returns (5 x N float32 rows, N x bad_bits / 8 bytes or None,
    per-level statistics).  Output order: levels ascending, canonical order inside a level (spec S1).  hashsift: instead of BAD
    bytes, the HashSIFT 129-vectors of the keypoints (hash_sift.cpp on the blurred level, cropping scale 1: createDescriber, .cpp:58-62)."""
    geo = pyramid_geometry(img.shape[0], img.shape[1], scale_factor, nlevels)
    quota = level_quotas(nfeatures, scale_factor, nlevels)
    level = img
    rows_out, desc_out, stats = [], [], []
    for s, (h, w, scale) in enumerate(geo):
        if s > 0:
            level = resize_linear(level, h, w)
        tiles_x = (w + TILE - 1) // TILE
        ys, xs = np.nonzero(fast9(level, fast_threshold))
        order = canonical_order(xs, ys, tiles_x)
        xs, ys = xs[order], ys[order]
        ncand = len(xs)
        cap = cv_round(CORNER_DENSITY * (h * w))                            # .cpp:252; spec S2: the first `cap` in canonical order
        xs, ys = xs[:cap], ys[:cap]
        resp = harris(level, xs, ys) if len(xs) else np.zeros(0, np.float32)
        keep = radius_suppression(xs, ys, resp, nonmax_radius) if len(xs) else np.zeros(0, bool)
        xs, ys, resp = xs[keep], ys[keep], resp[keep]
        nsurv = len(xs)
        if nsurv > quota[s]:                                                # limitPoints (.cu:344-358) under spec S3
            pick = np.lexsort((xs, ys, -resp.astype(np.float64)))[:quota[s]]
            pick = pick[canonical_order(xs[pick], ys[pick], tiles_x)]       # the quota filters, the canonical order stays
            xs, ys, resp = xs[pick], ys[pick], resp[pick]
        n = len(xs)
        stats.append(dict(candidates=ncand, after_cap=min(ncand, cap), after_nms=nsurv, kept=n))
        if n == 0:
            continue
        ang = np.array([ic_angle(level, int(x), int(y))[0] for x, y in zip(xs, ys)], np.float32)
        if hashsift:
            kp = np.stack([xs.astype(np.float32), ys.astype(np.float32), np.full(n, PATCH_SIZE, np.float32), ang], 1)
            desc_out.append(so.hashsift_vectors(gaussian7(level), kp, 1.0)[0])
        elif bad_bits:
            blur = gaussian7(level)                                         # .cpp:305: the describer sees the blurred level,
            kp = np.stack([xs.astype(np.float32), ys.astype(np.float32), np.full(n, PATCH_SIZE, np.float32), ang], 1)   # convertKeypoints: size 31
            desc_out.append(so.bad_describe(blur, kp, bad_bits, scale_factor=1.0))      # createDescriber: BAD::create(1, ..) (.cpp:53-56)
        # scalePoints (.cu:236-248): (short)(scale * x + 0.5f), mul and add rounded separately (spec S8)
        ox = np.trunc((scale * xs.astype(np.float32)).astype(np.float32) + f32(0.5)).astype(np.int16)
        oy = np.trunc((scale * ys.astype(np.float32)).astype(np.float32) + f32(0.5)).astype(np.int16)
        r = np.zeros((5, n), np.float32)
        r[0] = (ox.astype(np.uint16).astype(np.uint32) | (oy.astype(np.uint16).astype(np.uint32) << 16)).view(np.float32)      # short2 in 4 bytes
        r[1] = resp
        r[2] = ang
        r[3] = np.full(n, s, np.int32).view(np.float32)
        r[4] = f32(scale * f32(PATCH_SIZE))
        rows_out.append(r)
    kps = np.concatenate(rows_out, axis=1) if rows_out else np.zeros((5, 0), np.float32)
    if hashsift:
        desc = np.concatenate(desc_out, axis=0) if desc_out else np.zeros((0, 129), np.float32)
    else:
        desc = (np.concatenate(desc_out, axis=0) if desc_out else np.zeros((0, bad_bits // 8), np.uint8)) if bad_bits else None
    return kps, desc, stats
