"""HashSIFT fuzz (test_fuzz_compute) seeds whose 129-vectors differ from the reference arithmetic: how the differing elements spread
over the keypoints (rows), and whether the keypoints concerned see the same patch.  usage: python tools/microbench/hs_vec_case.py seed ..."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
import cef_loader
from oracle import pyoracle as O
import test_gpu_fuzz as F
cef = cef_loader.load()
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(10_000 + seed)
    rows, cols = int(rng.integers(20, 300)), int(rng.integers(20, 400))
    kindimg = int(rng.integers(0, 5))
    rng = np.random.default_rng(10_000 + seed); rows, cols = int(rng.integers(20, 300)), int(rng.integers(20, 400))
    img = F._natural(seed, rows, cols, F._image(rng, rows, cols, int(rng.integers(0, 5))))
    n = int(rng.integers(1, 300))
    kps = np.zeros((n, 4), np.float32)
    kps[:, 0] = rng.uniform(-20, cols + 20, n); kps[:, 1] = rng.uniform(-20, rows + 20, n)
    kps[:, 2] = rng.choice([31.0, 31.0, 7.0, 12.5, 48.0, 64.0, 90.0], n); kps[:, 3] = rng.uniform(0, 360, n)
    sel = rng.random(n); kps[sel < 0.1, 3] = -1.0; kps[(sel >= 0.1) & (sel < 0.15), 3] = -7.0
    if rng.random() < 0.5: kps[:, :2] = np.floor(kps[:, :2])
    scale = float(rng.choice([1.0, 1.0, 0.75, 1.5, 2.0]))
    kps[:, 2] = np.minimum(kps[:, 2], np.float32(110.0 / scale))
    hs = cef.HashSIFT.create(scale, cef.HashSIFT.SIZE_256_BITS)
    resp, _ = hs.debug(torch.from_numpy(img).cuda(), torch.from_numpy(kps).cuda(), max_size=float(kps[:, 2].max()))
    torch.cuda.synchronize()
    resp = resp.cpu().numpy()
    want = O.hashsift_responses(img, kps, crop_scale=scale)
    d = resp - want
    rows_off = np.nonzero((d != 0).any(axis=1))[0]
    print(f"seed {seed}: image kind {kindimg} natural {seed % 7 in (5, 6)} {img.shape} n {n} scale {scale}: {np.count_nonzero(d)} elements differ in {len(rows_off)} vectors; "
          f"distinct expected vectors {len(np.unique(want, axis=0))} of {n}; image values {np.unique(img).size}")
    for r in rows_off[:6]:
        nz = np.nonzero(d[r])[0]
        print("   keypoint", r, kps[r], "elements", nz[:10], "diffs", d[r][nz[:10]], "vector max", want[r].max(), "nonzero", np.count_nonzero(want[r]))
