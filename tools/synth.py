"""Seeded synthetic frames for tests and bench (SURVEY 8d "Synthetic inputs").

A corner-rich u8 frame: random filled rectangles and triangles at several scales on a mid-gray
background plus low-amplitude noise.  Frame k of the benchmark uses seed 1000+k.  Pure numpy
(default_rng / PCG64), so the same frame is produced in the build container and on the GPU box.
"""
import numpy as np

SIZES = {"vga": (480, 640), "720p": (720, 1280), "fhd": (1080, 1920), "2.7k": (1520, 2704), "4k": (2160, 3840), "8k": (4320, 7680)}


def synth_frame(rows, cols, seed=1000, density=0.3, noise=3):
    rng = np.random.default_rng(seed)
    img = np.full((rows, cols), 128, dtype=np.int16)
    # shapes per megapixel, tuned so that every pyramid level has more radius-15 NMS survivors than its
    # quota at nfeatures=40000 while FAST candidates stay far below the 10% cap
    nshapes = int(density * 2600 * rows * cols / 1e6)
    scales = rng.choice([10, 18, 32, 56, 96], size=nshapes, p=[0.34, 0.28, 0.2, 0.12, 0.06])
    cx = rng.integers(0, cols, size=nshapes)
    cy = rng.integers(0, rows, size=nshapes)
    w = np.maximum(4, (scales * rng.uniform(0.5, 1.5, size=nshapes)).astype(np.int64))
    h = np.maximum(4, (scales * rng.uniform(0.5, 1.5, size=nshapes)).astype(np.int64))
    val = rng.integers(0, 256, size=nshapes)
    kind = rng.random(nshapes) < 0.3          # True -> triangle
    tri = rng.uniform(0.0, 1.0, size=(nshapes, 6))
    for i in range(nshapes):
        x0 = max(0, cx[i] - w[i] // 2); x1 = min(cols, cx[i] + (w[i] + 1) // 2)
        y0 = max(0, cy[i] - h[i] // 2); y1 = min(rows, cy[i] + (h[i] + 1) // 2)
        if x1 <= x0 or y1 <= y0:
            continue
        if not kind[i]:
            img[y0:y1, x0:x1] = val[i]
        else:
            # triangle with vertices inside the bounding box (half-plane test)
            vx = x0 + tri[i, 0:3] * (x1 - x0)
            vy = y0 + tri[i, 3:6] * (y1 - y0)
            yy, xx = np.mgrid[y0:y1, x0:x1]
            d0 = (vx[1] - vx[0]) * (yy - vy[0]) - (vy[1] - vy[0]) * (xx - vx[0])
            d1 = (vx[2] - vx[1]) * (yy - vy[1]) - (vy[2] - vy[1]) * (xx - vx[1])
            d2 = (vx[0] - vx[2]) * (yy - vy[2]) - (vy[0] - vy[2]) * (xx - vx[2])
            inside = ((d0 >= 0) & (d1 >= 0) & (d2 >= 0)) | ((d0 <= 0) & (d1 <= 0) & (d2 <= 0))
            sub = img[y0:y1, x0:x1]
            sub[inside] = val[i]
    if noise > 0:
        img += rng.integers(-noise, noise + 1, size=img.shape, dtype=np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


def noise_frame(rows, cols, seed=7):
    """Uniform noise: > 10% FAST corners, exercises the candidate cap (spec S2)."""
    return np.random.default_rng(seed).integers(0, 256, size=(rows, cols), dtype=np.uint8)


def random_keypoints(rows, cols, n, seed=3, size=31.0, border=0.0, special=True):
    """(n,4) float32 {x, y, size, angle}: integer-ish and sub-pixel positions, all orientations, and (if
    special) the angle == -1 / angle < 0 / near-border cases the reference branches on (bad.cpp:127,138)."""
    rng = np.random.default_rng(seed)
    k = np.zeros((n, 4), dtype=np.float32)
    k[:, 0] = rng.uniform(border, cols - 1 - border, size=n)
    k[:, 1] = rng.uniform(border, rows - 1 - border, size=n)
    half = n // 2
    k[:half, 0:2] = np.floor(k[:half, 0:2])
    k[:, 2] = size
    k[:, 3] = rng.uniform(0, 360, size=n)
    if special and n >= 16:
        k[0::16, 3] = -1.0        # axis-aligned branch
        k[1::16, 3] = -37.5       # angle < 0, != -1: cos=1 sin=0
        k[2::16, 3] = 0.0
        k[3::16, 0] = rng.uniform(0, 20, size=k[3::16].shape[0])              # left border
        k[4::16, 1] = rng.uniform(rows - 21, rows - 1, size=k[4::16].shape[0])  # bottom border
    return k
