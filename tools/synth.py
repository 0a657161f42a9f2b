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


def _bilinear_up(g, rows, cols, cell):
    """Bilinear interpolation of a coarse grid (float64, elementwise IEEE operations only: the same frame on every host)."""
    y = np.arange(rows, dtype=np.float64) / cell
    x = np.arange(cols, dtype=np.float64) / cell
    y0 = np.floor(y).astype(np.int64); x0 = np.floor(x).astype(np.int64)
    ty = (y - y0)[:, None]; tx = (x - x0)[None, :]
    a = g[np.ix_(y0, x0)]; b = g[np.ix_(y0, x0 + 1)]; c = g[np.ix_(y0 + 1, x0)]; d = g[np.ix_(y0 + 1, x0 + 1)]
    return (a * (1 - tx) + b * tx) * (1 - ty) + (c * (1 - tx) + d * tx) * ty


def powerlaw_frame(rows, cols, seed=1000, beta=1.3, contrast=45.0):
    """1/f-textured frame: the amplitude spectrum of natural photographs falls off as 1/f^beta.  Built as
    octaves of bilinearly interpolated Gaussian noise, cell size c weighted c^(beta - 1) (beta = 1: the same variance in every
    octave, which is what a 1/f amplitude spectrum means in two dimensions), down to single pixels -- texture at every
    scale, soft gradients, no flat areas: the regime of FAST threshold ties, Harris near-ties and HashSIFT projections close
    to zero that rectangles-and-noise frames do not reach.  beta 1.3 (photographs: 1.0 .. 1.5) gives ~3 % FAST corners at
    threshold 20, beta 1.0 ~17 % (the 10 % cap becomes active)."""
    rng = np.random.default_rng(seed)
    acc = np.zeros((rows, cols), np.float64)
    c = 1
    while c <= max(8, min(rows, cols) // 2):
        g = rng.standard_normal((rows // c + 2, cols // c + 2))
        acc += (float(c) ** (beta - 1.0)) * (_bilinear_up(g, rows, cols, c) if c > 1 else g[:rows, :cols])
        c *= 2
    acc -= acc.mean()
    s = acc.std()
    acc = 128.0 + acc * (contrast / s if s > 0 else 0.0)
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def powerlaw_frames_tiled(rows, cols, seed=1000, betas=(1.3, 1.0), contrast=45.0, tile_div=2):
    """Frames with the statistics of powerlaw_frame for the bench line's data-dependence rows: the octave fields are generated
    ONCE at (rows / tile_div, cols / tile_div) for all `betas` (the interpolation of the octaves is what costs: 34 s per 8K frame
    otherwise) and mirrored into the full frame (continuous across the seams: no artificial edges).  Not bit-identical to
    powerlaw_frame -- nothing is compared against these frames but the oracle run on the very same array."""
    r, c_ = rows // tile_div, cols // tile_div
    rng = np.random.default_rng(seed)
    accs = [np.zeros((r, c_), np.float64) for _ in betas]
    c = 1
    while c <= max(8, min(r, c_) // 2):
        g = rng.standard_normal((r // c + 2, c_ // c + 2))
        f = _bilinear_up(g, r, c_, c) if c > 1 else g[:r, :c_]
        for a, b in zip(accs, betas):
            a += (float(c) ** (b - 1.0)) * f
        c *= 2
    out = []
    for a in accs:
        a -= a.mean()
        sd = a.std()
        t = np.clip(np.rint(128.0 + a * (contrast / sd if sd > 0 else 0.0)), 0, 255).astype(np.uint8)
        row = np.concatenate([t if k % 2 == 0 else t[:, ::-1] for k in range(tile_div)], axis=1)
        full = np.concatenate([row if k % 2 == 0 else row[::-1, :] for k in range(tile_div)], axis=0)
        out.append(np.ascontiguousarray(full[:rows, :cols]))
    return out


def _gauss_blur(img, sigma):
    r = max(1, int(np.ceil(3 * sigma)))
    k = np.exp(-0.5 * (np.arange(-r, r + 1, dtype=np.float64) / sigma) ** 2)
    k /= k.sum()
    p = np.pad(img, ((0, 0), (r, r)), mode="edge")
    t = np.zeros(img.shape, np.float64)
    for j in range(2 * r + 1):
        t += k[j] * p[:, j:j + img.shape[1]]
    p = np.pad(t, ((r, r), (0, 0)), mode="edge")
    o = np.zeros(img.shape, np.float64)
    for j in range(2 * r + 1):
        o += k[j] * p[j:j + img.shape[0], :]
    return o


def blurred_edges_frame(rows, cols, seed=1000, density=0.6, sigmas=(0.6, 1.2, 2.5, 5.0)):
    """The shapes of synth_frame seen through defocus: vertical bands of the frame are blurred with
    Gaussians of several widths (band k with sigmas[k]), then sensor noise is added -- edges a few pixels wide, corners whose
    FAST arcs sit at the threshold, the gradient statistics of photographs of man-made scenes."""
    base = synth_frame(rows, cols, seed=seed, density=density, noise=0).astype(np.float64)
    out = np.empty_like(base)
    nb = len(sigmas)
    edges = [cols * k // nb for k in range(nb + 1)]
    for k, s in enumerate(sigmas):
        x0, x1 = edges[k], edges[k + 1]
        if x1 <= x0:
            continue
        m = int(np.ceil(3 * s)) + 1
        a, b = max(0, x0 - m), min(cols, x1 + m)
        out[:, x0:x1] = _gauss_blur(base[:, a:b], s)[:, x0 - a:x0 - a + (x1 - x0)]
    rng = np.random.default_rng(seed + 99991)
    out += rng.integers(-2, 3, size=out.shape)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)
