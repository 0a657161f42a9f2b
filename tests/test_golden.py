"""Golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py).

descriptors_probe.npz holds outputs of the REFERENCE CPU descriptors (their FNV hashes are the ones SURVEY.md
Appendix B recorded from modules/efficient_features/src/bad.cpp / hash_sift.cpp); detector_*.npz hold outputs of the
spec-defining CPU restatement (the reference has no CPU detector; parity unpinned there).  The CPU half checks
the oracle against them, the GPU half (-m gpu) checks the HIP path through the C ABI."""
import glob
import os

import numpy as np
import pytest

from tests.lcg_probe import REFERENCE_HASHES, fnv1a32

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DETECTOR_FILES = sorted(glob.glob(os.path.join(HERE, "detector_*.npz")))
# must mirror tests/golden/make_golden.py DETECTOR_CASES
DETECTOR_KW = {"synth_240x320": dict(nfeatures=1500), "synth_480x640": dict(nfeatures=4000),
               "synth_480x640_r5_t40": dict(nfeatures=4000, nonmax_radius=5, fast_threshold=40),
               "noise_200x260_cap": dict(nfeatures=3000), "squares_200x260_ties": dict(nfeatures=2000)}
DESC_TAGS = ["bad256", "bad512", "hashsift256", "hashsift512"]


def hashsift_byte_tolerance(nbytes_per_desc, total_bytes):
    """HashSIFT bytes HIP-vs-CPU: the reference's own GPU-vs-CPU tolerance is 1e-4 of the bytes
    (tests/descriptor_test.cpp:72).  On samples of a few hundred descriptors that rounds to less than the effect of ONE
    129-vector element differing by one unit (fixed-point vs sequentially rounded float histogram sums, measured rate
    1.5e-6 per element), which moves every projection T by one weight and flips about one bit per 256: allow that."""
    return max(nbytes_per_desc // 32, int(1e-4 * total_bytes))


def _case(path):
    name = os.path.basename(path)[len("detector_"):-len(".npz")]
    return name, np.load(path), DETECTOR_KW[name]


def test_fixture_inventory():
    assert len(DETECTOR_FILES) == len(DETECTOR_KW) and os.path.exists(os.path.join(HERE, "descriptors_probe.npz"))


def test_probe_fixture_is_reference_output():
    g = np.load(os.path.join(HERE, "descriptors_probe.npz"))
    for kind in ("bad", "hashsift"):
        for nbits in (256, 512):
            assert fnv1a32(g[f"{kind}{nbits}"]) == REFERENCE_HASHES[(kind, nbits)]


def test_oracle_descriptors_equal_probe_fixture(oracle):
    g = np.load(os.path.join(HERE, "descriptors_probe.npz"))
    for nbits in (256, 512):
        assert np.array_equal(oracle.bad_compute(g["image"], g["keypoints"], nbits), g[f"bad{nbits}"])
        assert np.array_equal(oracle.hashsift_compute(g["image"], g["keypoints"], nbits), g[f"hashsift{nbits}"])


@pytest.mark.parametrize("path", DETECTOR_FILES, ids=[os.path.basename(p)[9:-4] for p in DETECTOR_FILES])
def test_oracle_detector_equals_fixture(oracle, path):
    name, g, kw = _case(path)
    img = g["image"]
    assert np.array_equal(oracle.pyramid_level(img, 1), g["level1"])
    assert np.array_equal(oracle.gaussian7(g["level1"]), g["level1_blur"])
    assert np.array_equal(oracle.fast9_detect(img, threshold=kw.get("fast_threshold", 20), border=15), g["fast_l0"])
    for dt, tag in enumerate(DESC_TAGS):
        r = oracle.detect_and_compute(img, desc_type=dt, **kw)
        assert np.array_equal(r["kps"].view(np.uint32), g["kps"])
        assert np.array_equal(r["lvl_xy"], g["lvl_xy"])
        assert np.array_equal(r["desc"], g[tag])
        for k in ("n_candidates", "n_after_cap", "n_after_nms", "n_kept"):
            assert list(g[k]) == r["stats"][k]
    if "cap" in name:
        assert (g["n_candidates"] > g["n_after_cap"]).any()


# ---------------------------------------------------------------- GPU half
@pytest.fixture(scope="module")
def cef():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cef_loader
    return cef_loader.load()


@pytest.mark.gpu
def test_hip_descriptors_equal_probe_fixture(cef):
    g = np.load(os.path.join(HERE, "descriptors_probe.npz"))
    for nbits, enum in ((256, cef.BAD.SIZE_256_BITS), (512, cef.BAD.SIZE_512_BITS)):
        assert np.array_equal(cef.BAD.create(1.0, enum).compute(g["image"], g["keypoints"]), g[f"bad{nbits}"])
    for nbits, enum in ((256, cef.HashSIFT.SIZE_256_BITS), (512, cef.HashSIFT.SIZE_512_BITS)):
        got = cef.HashSIFT.create(1.0, enum).compute(g["image"], g["keypoints"])
        assert np.count_nonzero(got != g[f"hashsift{nbits}"]) <= hashsift_byte_tolerance(nbits // 8, got.size)


@pytest.mark.gpu
@pytest.mark.parametrize("path", DETECTOR_FILES, ids=[os.path.basename(p)[9:-4] for p in DETECTOR_FILES])
def test_hip_detector_equals_fixture(cef, path):
    import torch
    name, g, kw = _case(path)
    d_img = torch.from_numpy(g["image"]).cuda()
    for dt, tag in enumerate(DESC_TAGS):
        det = cef.EfficientFeatures.create(kw["nfeatures"], 1.2, 8, 0, kw.get("fast_threshold", 20), kw.get("nonmax_radius", 15), dt)
        kps, desc, cnt = det.detectAndComputeAsync(d_img)
        torch.cuda.synchronize()
        n = int(cnt.item())
        assert n == g["kps"].shape[1]
        assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), g["kps"])       # bit-exact 5xN matrix
        got = desc[:n].cpu().numpy()
        if tag.startswith("bad"):
            assert np.array_equal(got, g[tag])
        else:
            assert np.count_nonzero(got != g[tag]) <= hashsift_byte_tolerance(got.shape[1], got.size)
        st = det.lastLevelStats()
        assert [s["n_candidates"] for s in st] == list(g["n_candidates"])
        assert [s["n_kept"] for s in st] == list(g["n_kept"])
    # intermediate stage: pyramid level 1 as the device computed it
    lvl1 = det.copyLevel(1, g["image"].shape[0], g["image"].shape[1])
    torch.cuda.synchronize()
    assert np.array_equal(lvl1.cpu().numpy(), g["level1"])
