"""Seeded fuzz parity: random frame sizes, image kinds, detector parameters, masks and descriptor types, HIP path vs the
oracle through the C-ABI.  Bit-exact keypoints (location, response bits, angle bits, octave, size) and BAD bytes;
HashSIFT within the byte tolerance of tests/test_golden.py (spec S8).  The cases are small so the oracle stays fast;
every case prints its parameters on failure, so a failing seed is a ready-made regression test."""
import numpy as np
import pytest

import cef_loader
from tools import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cef():
    return cef_loader.load()


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    return torch


def _image(rng, rows, cols, kind):
    if kind == 0:
        return synth.synth_frame(rows, cols, seed=int(rng.integers(1 << 30)), density=float(rng.choice([0.3, 1.0, 3.0])))
    if kind == 1:
        return synth.noise_frame(rows, cols, seed=int(rng.integers(1 << 30)))
    if kind == 2:                                   # smooth ramps + a few hard edges: long runs of equal responses (tie rules)
        y, x = np.mgrid[0:rows, 0:cols]
        img = ((x * 3 + y * 5) & 255).astype(np.uint8)
        img[rows // 3: rows // 3 + 7, :] = 255
        img[:, cols // 2: cols // 2 + 5] = 0
        return img
    if kind == 3:                                   # checkerboard of random period: a corner at every crossing
        p = int(rng.integers(3, 24))
        y, x = np.mgrid[0:rows, 0:cols]
        return ((((x // p) + (y // p)) & 1) * int(rng.integers(40, 256))).astype(np.uint8)
    # sparse bright dots on black
    img = np.zeros((rows, cols), np.uint8)
    n = int(rng.integers(1, 200))
    img[rng.integers(0, rows, n), rng.integers(0, cols, n)] = rng.integers(60, 256, n)
    return img


def _case(seed):
    rng = np.random.default_rng(seed)
    rows, cols = int(rng.integers(33, 420)), int(rng.integers(33, 520))
    kw = dict(nfeatures=int(rng.choice([1, 7, 100, 1000, 5000, 20000])),
              scale_factor=float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.5, 2.0])),
              nlevels=int(rng.integers(1, 9)),
              first_level=0,
              fast_threshold=int(rng.choice([1, 5, 10, 20, 20, 40, 90])),
              nonmax_radius=int(rng.choice([0, 1, 3, 8, 15, 15, 15, 16, 17, 24, 33])))
    img = _image(rng, rows, cols, int(rng.integers(0, 5)))
    mask = None
    if rng.random() < 0.3:
        mask = (rng.random((rows, cols)) < 0.7).astype(np.uint8) * 255
        mask[: rows // 4, : cols // 3] = 0
    desc_type = int(rng.choice([-1, 0, 1, 1, 2, 3]))
    return img, mask, desc_type, kw


def _hashsift_tol(nbits, n):
    return 2 * max(1, n // 100)


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_detect_and_compute(cef, torch_mod, oracle, seed):
    torch = torch_mod
    img, mask, desc_type, kw = _case(seed)
    info = f"seed {seed}: {img.shape} desc_type {desc_type} mask {mask is not None} {kw}"
    det = cef.EfficientFeatures.create(kw["nfeatures"], kw["scale_factor"], kw["nlevels"], kw["first_level"],
                                       kw["fast_threshold"], kw["nonmax_radius"], max(desc_type, 0))
    d_img = torch.from_numpy(img).cuda()
    d_mask = None if mask is None else torch.from_numpy(mask).cuda()
    if desc_type >= 0:
        kps, desc, cnt = det.detectAndComputeAsync(d_img, mask=d_mask)
    elif d_mask is not None:
        kps, desc, cnt = det.detectAndComputeAsync(d_img, mask=d_mask, want_descriptors=False)
    else:
        kps, cnt = det.detectAsync(d_img)
        desc = None
    torch.cuda.synchronize()
    n = int(cnt.item())
    ref = oracle.detect_and_compute(img, desc_type=desc_type, mask=mask, **kw)
    st = det.lastLevelStats()
    for l, s in enumerate(st):
        assert s["n_candidates"] == ref["stats"]["n_candidates"][l], f"{info}: level {l} FAST corners"
        assert s["n_after_nms"] == ref["stats"]["n_after_nms"][l], f"{info}: level {l} NMS survivors"
        assert s["n_kept"] == ref["stats"]["n_kept"][l], f"{info}: level {l} kept"
    assert n == ref["n"], info
    g, r = kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32)
    assert np.array_equal(g, r), f"{info}: keypoint rows differ at {np.argwhere(g != r)[:4].tolist()}"
    if desc_type in (0, 1):
        assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"]), info
    elif desc_type in (2, 3):
        d = desc[:n].cpu().numpy()
        assert d.shape == ref["desc"].shape, info
        nbad = int(np.count_nonzero(d != ref["desc"]))
        assert nbad <= _hashsift_tol(256 if desc_type == 2 else 512, max(n, 1)), f"{info}: {nbad} descriptor bytes differ"
