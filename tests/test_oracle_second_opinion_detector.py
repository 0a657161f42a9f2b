"""The C oracle's detectAndCompute against a SECOND restatement of the whole flow (tests/second_opinion_detector.py: numpy, brute
force, written from the reference text and DESIGN.md section 3, not from oracle/efx_oracle.c).  Like tests/test_oracle_second_opinion.py
this pins nothing on the reference (no OpenCV, no CUDA here: the detector is parity-unpinned and stays so); it removes the
single-transcriber risk for what had no second opinion yet -- level geometry and quotas, the 10 % cap, the suppression predicate
with its ties, the quota's order, the canonical output order, the moments, scalePoints, the row encoding, and "describe on the
blurred level at level coordinates".  Everything must be equal bit for bit: keypoint rows (x, y, response, angle, octave, size) and
BAD bytes."""
import numpy as np
import pytest

from tests import second_opinion_detector as sod
from tools import synth

BAD = {256: 0, 512: 1}           # EfficientFeatures::DescriptorType (cuda_efficient_features.h:39-45)


def _compare(oracle, img, nfeatures, nlevels, bits, **kw):
    want_k, want_d, stats = sod.detect_and_compute(img, nfeatures, nlevels=nlevels, bad_bits=bits, **kw)
    got = oracle.detect_and_compute(img, nfeatures=nfeatures, nlevels=nlevels, desc_type=BAD[bits] if bits else -1,
                                    scale_factor=kw.get("scale_factor", 1.2), fast_threshold=kw.get("fast_threshold", 20),
                                    nonmax_radius=kw.get("nonmax_radius", 15))
    for lvl, st in enumerate(stats):
        assert got["stats"]["n_candidates"][lvl] == st["candidates"], f"level {lvl}: FAST corners"
        assert got["stats"]["n_after_cap"][lvl] == st["after_cap"], f"level {lvl}: cap"
        assert got["stats"]["n_after_nms"][lvl] == st["after_nms"], f"level {lvl}: suppression"
        assert got["stats"]["n_kept"][lvl] == st["kept"], f"level {lvl}: quota"
    assert got["kps"].shape == want_k.shape
    for row, name in enumerate(("xy", "response", "angle", "octave", "size")):
        assert np.array_equal(got["kps"][row].view(np.uint32), want_k[row].view(np.uint32)), name
    if bits:
        assert np.array_equal(got["desc"], want_d)
    return stats


def test_geometry_and_quotas_match(oracle):
    for rows, cols, sf, nl in ((1080, 1920, 1.2, 8), (2160, 3840, 1.2, 8), (4320, 7680, 1.2, 8), (481, 643, 1.37, 5), (300, 400, 2.0, 3)):
        lr, lc, sc = oracle.pyramid_geometry(rows, cols, scale_factor=sf, nlevels=nl)
        want = sod.pyramid_geometry(rows, cols, sf, nl)
        assert [(int(a), int(b)) for a, b in zip(lr, lc)] == [(h, w) for h, w, _ in want]
        assert np.array_equal(np.asarray(sc, np.float32), np.array([s for _, _, s in want], np.float32))
        for total in (500, 5000, 40000, 7):
            assert list(oracle.level_quotas(total, scale_factor=sf, nlevels=nl)) == sod.level_quotas(total, sf, nl)


@pytest.mark.parametrize("bits", [256, 512])
def test_whole_flow_quota_active(oracle, bits):
    """Every level has more survivors than its quota: the (response, y, x) cut and the canonical order behind it."""
    img = synth.synth_frame(300, 420, seed=31, density=2.0)
    stats = _compare(oracle, img, nfeatures=150, nlevels=4, bits=bits)
    assert all(st["after_nms"] > st["kept"] for st in stats[:3])


def test_whole_flow_quota_inactive_and_detect_only(oracle):
    img = synth.synth_frame(260, 380, seed=32)
    stats = _compare(oracle, img, nfeatures=100000, nlevels=5, bits=0)
    assert all(st["after_nms"] == st["kept"] for st in stats)
    _compare(oracle, img, nfeatures=100000, nlevels=5, bits=256)


def test_whole_flow_cap_active_and_ties(oracle):
    """Noise: FAST fires on more than 10 % of the pixels (the cap cuts in canonical order, S2); a two-valued checker pattern
    gives equal Harris responses within the radius (`<=` suppresses both, .cu:90)."""
    noise = synth.noise_frame(160, 200, seed=5)
    stats = _compare(oracle, noise, nfeatures=400, nlevels=2, bits=256, fast_threshold=10)
    assert stats[0]["candidates"] > stats[0]["after_cap"]
    yy, xx = np.mgrid[0:200, 0:260]
    checker = (((yy // 9) + (xx // 9)) % 2 * 200 + 20).astype(np.uint8)
    _compare(oracle, checker, nfeatures=5000, nlevels=3, bits=512)


def test_whole_flow_other_parameters(oracle):
    img = synth.powerlaw_frame(280, 360, seed=9, beta=1.1, contrast=70.0)
    _compare(oracle, img, nfeatures=300, nlevels=3, bits=256, scale_factor=1.5, fast_threshold=12, nonmax_radius=7)
    _compare(oracle, img, nfeatures=300, nlevels=6, bits=0, scale_factor=1.1, fast_threshold=30, nonmax_radius=20)


@pytest.mark.parametrize("nbits", [256, 512])
def test_whole_flow_hashsift(oracle, nbits):
    """detectAndCompute with a HashSIFT describer: the oracle's descriptor bytes against the second restatement's 129-vectors of the
    same keypoints on the blurred level (tests/second_opinion.py), projected in double -- bits equal wherever the projection is
    not within float rounding of zero."""
    from tests import second_opinion as so
    img = synth.synth_frame(280, 380, seed=41, density=1.5)
    _, vec, _ = sod.detect_and_compute(img, 200, nlevels=3, hashsift=True)
    got = oracle.detect_and_compute(img, nfeatures=200, nlevels=3, desc_type=2 if nbits == 256 else 3)
    assert got["n"] == len(vec) > 100
    want, T = so.hashsift_bits(vec, nbits)
    decided = np.abs(T) > 1e-3
    bg = np.unpackbits(got["desc"], axis=1).astype(bool)
    bw = np.unpackbits(want, axis=1).astype(bool)
    assert np.array_equal(bg[decided], bw[decided])
    assert decided.mean() > 0.999
