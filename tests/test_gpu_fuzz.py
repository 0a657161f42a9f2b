"""Seeded fuzz parity: random frame sizes, image kinds (shapes, noise, ramps, checkerboards, dots, 1/f texture, defocused edges), detector parameters, masks and descriptor types, HIP path vs the
oracle through the C-ABI.  Bit-exact keypoints (location, response bits, angle bits, octave, size) and BAD bytes;
HashSIFT within the byte tolerance of tests/test_golden.py (spec S8).  The cases are small so the oracle stays fast;
every case prints its parameters on failure, so a failing seed is a ready-made regression test."""
import os

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


def _natural(seed, rows, cols, img):
    """Seeds 5 and 6 modulo 7 replace the drawn image by one with natural-image statistics, from a
    generator of their own, so that every other draw of the case -- and every other seed -- stays what earlier sweeps ran."""
    if seed % 7 == 5:
        r2 = np.random.default_rng(seed + 777_000)
        return synth.powerlaw_frame(rows, cols, seed=seed, beta=float(r2.choice([1.0, 1.3, 1.6])), contrast=float(r2.choice([25.0, 45.0, 70.0])))
    if seed % 7 == 6:
        r2 = np.random.default_rng(seed + 777_000)
        sig = tuple(float(v) for v in r2.choice([0.5, 0.8, 1.2, 2.0, 3.5], size=int(r2.integers(1, 5))))
        return synth.blurred_edges_frame(rows, cols, seed=seed, density=float(r2.choice([0.6, 1.5, 3.0])), sigmas=sig)
    return img


def _case(seed):
    rng = np.random.default_rng(seed)
    rows, cols = int(rng.integers(33, 420)), int(rng.integers(33, 520))
    if rng.random() < 0.15:                         # some frames span many 64x64 tiles
        rows, cols = int(rng.integers(400, 1000)), int(rng.integers(500, 1400))
    kw = dict(nfeatures=int(rng.choice([1, 7, 100, 1000, 5000, 20000])),
              scale_factor=float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.5, 2.0])),
              nlevels=int(rng.integers(1, 9)),
              first_level=int(rng.choice([0, 0, 0, 0, 1, 2])),
              fast_threshold=int(rng.choice([1, 5, 10, 20, 20, 40, 90])),
              nonmax_radius=int(rng.choice([0, 1, 3, 8, 15, 15, 15, 16, 17, 24, 33])))
    img = _natural(seed, rows, cols, _image(rng, rows, cols, int(rng.integers(0, 5))))
    mask = None
    if rng.random() < 0.3:
        mask = (rng.random((rows, cols)) < 0.7).astype(np.uint8) * 255
        mask[: rows // 4, : cols // 3] = 0
    desc_type = int(rng.choice([-1, 0, 1, 1, 2, 3]))
    return img, mask, desc_type, kw


# HashSIFT tolerance.  The stated bound (the reference's own GPU-vs-CPU tolerance,
# tests/descriptor_test.cpp:72) is a FRACTION of the descriptor bytes of a whole data set: 1e-4.  It is asserted as such over
# everything this module compares (test_zz_hashsift_bytes_over_the_sweep).  A single case is small and its mismatches come in
# events -- a 129-vector element that differs from the CPU float sums by one unit (the device adds the norm's 128 squares in
# a tree, the CPU serially: section 3) flips the bits whose projection lies within one weight of zero, usually none, now and
# then three or four bytes of ONE descriptor -- so the per-case bound is the stated fraction plus ONE such event.  (Found by
# the round-3 sweep: seed 103188, 1935 keypoints x 64 bytes, 13 differing bytes against int(1e-4 x bytes) = 12.)
_HS_TOTAL = {"bytes": 0, "bad": 0, "elems": 0, "bad_elems": 0}


def _hashsift_check(nbad, nbits, n, info, got=None, want=None):
    nbytes = max(n, 0) * (nbits // 8)
    _HS_TOTAL["bytes"] += nbytes
    _HS_TOTAL["bad"] += nbad
    # every differing byte counts here: a localised defect -- one bad box or cell pattern repeated across the
    # keypoints of a case -- would show as far more than one rounding event per thousand bytes, whatever the de-duplication
    # below makes of it
    # (a periodic image -- a checkerboard -- holds the same patch under many keypoints, and one rounding event then repeats: the cap
    # scales with the repetition, keypoints per DISTINCT expected descriptor; found by the round-4 sweep, seed 915477: a 39 x 303
    # checkerboard, 490 keypoints, 104 differing bytes)
    rep = 1.0
    if want is not None and nbad and len(want):
        rep = len(want) / max(1, len(np.unique(want, axis=0)))
    # NOT the spec tolerance (1e-4 of the bytes, asserted below per distinct descriptor and sweep-wide at 4e-5): a sanity cap,
    # ten times looser on purpose, that catches localised defects before the de-duplication can hide them
    assert nbad <= (int(1e-3 * nbytes) + 4) * rep, (f"{info}: {nbad} of {nbytes} HashSIFT descriptor bytes differ (every byte counted, repetition {rep:.1f}; "
                                                     "sanity cap at 10x the tolerance, the tolerance itself is asserted below)")
    # A periodic image (checkerboards: kind 3) holds the same patch many times over, and ONE rounding event then shows up in
    # every keypoint that has it (found by the round-3 sweep: seed 705467, 24 keypoints of a checkerboard with the same two
    # bytes, 42 differing bytes against a bound of 21).  Events are counted once per distinct (expected, computed) descriptor;
    # the sweep-wide fraction below still counts every byte.
    if got is not None and nbad:
        pairs = np.unique(np.concatenate([want, got], axis=1), axis=0)
        half = want.shape[1]
        nbad = int(np.count_nonzero(pairs[:, :half] != pairs[:, half:]))
    assert nbad <= int(1e-4 * nbytes) + 4, f"{info}: {nbad} of {nbytes} HashSIFT descriptor bytes differ (distinct descriptors)"


def _hashsift_vectors_check(cef, oracle, img, kps, scale, info):
    """The 129-vectors themselves (not only the bytes they flip): the kernel must equal the CPU model of ITS OWN
    arithmetic bit for bit in every case, and differ from the reference's float arithmetic by at most one event per case
    (elements off by a few units; the rate over the sweep is asserted in test_zz_hashsift_bytes_over_the_sweep)."""
    import torch
    hs = cef.HashSIFT.create(scale, cef.HashSIFT.SIZE_256_BITS)
    resp, _ = hs.debug(torch.from_numpy(img).cuda(), torch.from_numpy(kps).cuda(), max_size=float(kps[:, 2].max()))
    torch.cuda.synchronize()
    resp = resp.cpu().numpy()
    model = oracle.hashsift_responses_fixedpoint(img, kps, crop_scale=scale)
    rows_off = np.nonzero((resp != model).any(axis=1))[0]
    assert rows_off.size == 0, f"{info}: {rows_off.size} vectors differ from the fixed-point model: {rows_off[:6].tolist()}"
    want = oracle.hashsift_responses(img, kps, crop_scale=scale)
    d = np.abs(resp - want)
    nbad = int(np.count_nonzero(d))
    _HS_TOTAL["elems"] += d.size
    _HS_TOTAL["bad_elems"] += nbad
    # One event = ONE vector whose normalisation lands on the other side of a rounding boundary: up to ~10 of its elements move
    # by one unit, all in the same direction (round-4 sweep, 60 000 cases: seeds 904066, 913349, 1000444 ... -- 7 to 10 elements
    # of a single vector each; tools/microbench/hs_vec_case.py).  Per case: few vectors touched, none by more than a few units;
    # the element RATE is asserted over the sweep.  (What guards the kernel itself is the bit-exact model comparison above.)
    assert d.max() <= 4.0, f"{info}: a 129-vector element is off by {d.max()}"
    nrows = int(np.count_nonzero((d != 0).any(axis=1)))
    assert nrows <= 3 + int(5e-3 * len(kps)), f"{info}: {nrows} of {len(kps)} 129-vectors differ from the reference arithmetic ({nbad} elements)"


# EFX_FUZZ_CASES / EFX_FUZZ_FIRST widen the sweep from the command line (the committed default keeps the suite fast)
_N = int(os.environ.get("EFX_FUZZ_CASES", "60"))
_FIRST = int(os.environ.get("EFX_FUZZ_FIRST", "0"))


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + _N))
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
    full = (f"{info}: gpu cand {[s['n_candidates'] for s in st]} nms {[s['n_after_nms'] for s in st]} kept {[s['n_kept'] for s in st]} n {n}"
            f" | oracle cand {list(ref['stats']['n_candidates'])} nms {list(ref['stats']['n_after_nms'])} kept {list(ref['stats']['n_kept'])} n {ref['n']}")
    for l, s in enumerate(st):
        assert s["n_candidates"] == ref["stats"]["n_candidates"][l], f"level {l} FAST corners: {full}"
        assert s["n_after_nms"] == ref["stats"]["n_after_nms"][l], f"level {l} NMS survivors: {full}"
        assert s["n_kept"] == ref["stats"]["n_kept"][l], f"level {l} kept: {full}"
    assert n == ref["n"], full
    g, r = kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32)
    assert np.array_equal(g, r), f"{info}: keypoint rows differ at {np.argwhere(g != r)[:4].tolist()}"
    if desc_type in (0, 1):
        assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"]), info
    elif desc_type in (2, 3):
        d = desc[:n].cpu().numpy()
        assert d.shape == ref["desc"].shape, info
        nbad = int(np.count_nonzero(d != ref["desc"]))
        _hashsift_check(nbad, 256 if desc_type == 2 else 512, n, info, d, ref["desc"])


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + max(_N // 2, 1)))
def test_fuzz_compute(cef, oracle, seed):
    """Stand-alone describers on random keypoints: positions inside / on the border / outside the frame, random sizes,
    every angle convention (-1 axis aligned, < 0 unrotated, degrees), random scale factors."""
    rng = np.random.default_rng(10_000 + seed)
    rows, cols = int(rng.integers(20, 300)), int(rng.integers(20, 400))
    img = _natural(seed, rows, cols, _image(rng, rows, cols, int(rng.integers(0, 5))))
    n = int(rng.integers(1, 300))
    kps = np.zeros((n, 4), np.float32)
    kps[:, 0] = rng.uniform(-20, cols + 20, n)
    kps[:, 1] = rng.uniform(-20, rows + 20, n)
    kps[:, 2] = rng.choice([31.0, 31.0, 7.0, 12.5, 48.0, 64.0, 90.0], n)
    kps[:, 3] = rng.uniform(0, 360, n)
    sel = rng.random(n)
    kps[sel < 0.1, 3] = -1.0
    kps[(sel >= 0.1) & (sel < 0.15), 3] = -7.0
    if rng.random() < 0.5:
        kps[:, :2] = np.floor(kps[:, :2])               # integer positions, as the detector reports them
    scale = float(rng.choice([1.0, 1.0, 0.75, 1.5, 2.0]))
    # a keypoint whose window exceeds the 160 KB LDS is refused (DESIGN.md limits; test_descriptor_keypoint_extremes)
    kps[:, 2] = np.minimum(kps[:, 2], np.float32(110.0 / scale))
    info = f"seed {seed}: {img.shape} n {n} scale {scale}"
    kind = int(rng.integers(0, 4))
    if kind < 2:
        nbits, enum = ((256, cef.BAD.SIZE_256_BITS), (512, cef.BAD.SIZE_512_BITS))[kind]
        got = cef.BAD.create(scale, enum).compute(img, kps)
        want = oracle.bad_compute(img, kps, nbits, scale_factor=scale)
        assert np.array_equal(got, want), f"{info} BAD{nbits}: rows {np.nonzero((got != want).any(axis=1))[0][:6].tolist()} differ"
    else:
        nbits, enum = ((256, cef.HashSIFT.SIZE_256_BITS), (512, cef.HashSIFT.SIZE_512_BITS))[kind - 2]
        got = cef.HashSIFT.create(scale, enum).compute(img, kps)
        want = oracle.hashsift_compute(img, kps, nbits, crop_scale=scale)
        nbad = int(np.count_nonzero(got != want))
        _hashsift_check(nbad, nbits, n, info, got, want)
        _hashsift_vectors_check(cef, oracle, img, kps, scale, info)


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + max(_N // 3, 1)))
def test_fuzz_provided_keypoints(cef, torch_mod, oracle, seed):
    """detectAndCompute(useProvidedKeypoints=True) on random raw 5 x N matrices (spec S13): positions anywhere on the frame,
    octaves inside and outside the pyramid, arbitrary angles."""
    torch = torch_mod
    rng = np.random.default_rng(20_000 + seed)
    rows, cols = int(rng.integers(40, 400)), int(rng.integers(40, 500))
    img = _natural(seed, rows, cols, _image(rng, rows, cols, int(rng.integers(0, 5))))
    nlevels = int(rng.integers(1, 9))
    scale = float(rng.choice([1.2, 1.2, 1.5, 2.0]))
    dt = int(rng.integers(0, 4))
    n = int(rng.integers(1, 200))
    kps = np.zeros((5, n), np.float32)
    x = rng.integers(0, cols, n).astype(np.uint32)
    y = rng.integers(0, rows, n).astype(np.uint32)
    kps[0] = (x | (y << 16)).view(np.float32)
    kps[1] = rng.random(n).astype(np.float32)
    kps[2] = rng.uniform(0, 360, n).astype(np.float32)
    octv = rng.integers(0, nlevels, n).astype(np.int32)
    octv[rng.random(n) < 0.05] = int(rng.choice([-1, nlevels, 31, 1000]))
    kps[3] = octv.view(np.float32)
    kps[4] = 31.0
    det = cef.EfficientFeatures.create(1000, scale, nlevels, 0, 20, 15, dt)
    got = det.detectAndComputeAsync(torch.from_numpy(img).cuda(), keypoints=torch.from_numpy(kps).cuda(), n=n,
                                    useProvidedKeypoints=True)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    want = oracle.compute_provided(img, kps, dt, scale_factor=scale, nlevels=nlevels)
    info = f"seed {seed}: {img.shape} n {n} levels {nlevels} scale {scale} desc_type {dt}"
    if dt <= 1:
        assert np.array_equal(got, want), f"{info}: rows {np.nonzero((got != want).any(axis=1))[0][:6].tolist()} differ"
    else:
        nbad = int(np.count_nonzero(got != want))
        _hashsift_check(nbad, 256 if dt == 2 else 512, n, info, got, want)


def test_every_small_frame_size(cef, torch_mod, oracle):
    """Frames from 1 x 1 upwards (every level geometry degenerates somewhere on the way): no crash, same result as the
    oracle -- including the single-launch pyramid, whose tiles and halos are mostly clipped at these sizes."""
    torch = torch_mod
    rng = np.random.default_rng(77)
    det = cef.EfficientFeatures.create(500, 1.2, 8, 0, 10, 3, cef.EfficientFeatures.BAD_256)
    for rows, cols in [(1, 1), (1, 50), (50, 1), (2, 2), (7, 9), (16, 16), (31, 31), (32, 33), (33, 32), (40, 47), (48, 64),
                       (63, 65), (64, 64), (65, 63), (96, 40), (40, 96), (100, 129), (129, 100)]:
        img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        img[rows // 3: rows // 3 + 9, cols // 3: cols // 3 + 9] = 255
        kps, desc, cnt = det.detectAndComputeAsync(torch.from_numpy(img).cuda())
        torch.cuda.synchronize()
        n = int(cnt.item())
        ref = oracle.detect_and_compute(img, desc_type=0, nfeatures=500, fast_threshold=10, nonmax_radius=3)
        assert n == ref["n"], (rows, cols)
        assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32)), (rows, cols)
        assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"]), (rows, cols)


def test_zz_hashsift_bytes_over_the_sweep():
    """The HashSIFT tolerance over everything the fuzz tests of this process compared.  Stated bound: at most 1e-4 of the
    descriptor bytes differ from the CPU reference arithmetic (descriptor_test.cpp:72).  Asserted: the MEASURED rates with the
    margin the C4 full-size test uses (3e-5 of the 129-vector elements, 4e-5 of the bytes; measured 1.4e-5 / 2e-5, DESIGN.md
    section 3), with a floor of two events for sweeps too small to resolve such rates."""
    if _HS_TOTAL["bytes"] == 0:
        pytest.skip("no HashSIFT case ran in this process")
    rate = _HS_TOTAL["bad"] / _HS_TOTAL["bytes"]
    erate = _HS_TOTAL["bad_elems"] / max(_HS_TOTAL["elems"], 1)
    print(f"\nHashSIFT over the sweep: {_HS_TOTAL['bad']} of {_HS_TOTAL['bytes']} bytes differ ({rate:.2e}); "
          f"{_HS_TOTAL['bad_elems']} of {_HS_TOTAL['elems']} 129-vector elements differ ({erate:.2e})")
    assert _HS_TOTAL["bad"] <= max(8, int(4e-5 * _HS_TOTAL["bytes"]))
    assert _HS_TOTAL["bad_elems"] <= max(8, int(3e-5 * _HS_TOTAL["elems"]))
