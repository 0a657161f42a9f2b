"""Pins the descriptor oracle on the only known-answer vectors that exist for this path: the FNV-1a-32
hashes of the reference CPU code's output recorded in SURVEY.md Appendix B (reference:
modules/efficient_features/src/bad.cpp:254-405, hash_sift.cpp:399-426)."""
import numpy as np
import pytest

from tests.lcg_probe import REFERENCE_HASHES, fnv1a32, probe_input


@pytest.mark.parametrize("nbits", [256, 512])
def test_bad_matches_reference_hash(oracle, nbits):
    img, kps = probe_input()
    desc = oracle.bad_compute(img, kps, nbits)
    assert desc.shape == (200, nbits // 8)
    assert fnv1a32(desc) == REFERENCE_HASHES[("bad", nbits)]


@pytest.mark.parametrize("nbits", [256, 512])
def test_hashsift_matches_reference_hash(oracle, nbits):
    img, kps = probe_input()
    desc = oracle.hashsift_compute(img, kps, nbits)
    assert desc.shape == (200, nbits // 8)
    assert fnv1a32(desc) == REFERENCE_HASHES[("hashsift", nbits)]


def test_probe_covers_border_and_interior(oracle):
    # the pinned vector exercises both BAD code paths (bad.cpp:345 border / :362 interior)
    img, kps = probe_input()
    b = 27.125
    border = (kps[:, 0] < b) | (kps[:, 0] + b >= img.shape[1]) | (kps[:, 1] < b) | (kps[:, 1] + b >= img.shape[0])
    assert border.sum() >= 10 and (~border).sum() >= 100


def test_fixed_point_histogram_model_vs_reference_order(oracle):
    """The HIP kernel sums the HashSIFT histogram in 32.32 fixed point (order independent); the CPU model of that
    arithmetic must stay within the stated tolerance of the reference's sequentially rounded float sums
    (hash_sift.cpp:233-290): at most 1e-4 of the 129-vector elements differ, by one unit (measured ~1.5e-6), and at
    most 1e-4 of the descriptor bytes (the reference's GPU-vs-CPU tolerance, tests/descriptor_test.cpp:72)."""
    from tools import synth
    img = synth.synth_frame(480, 640, seed=4)
    kps = synth.random_keypoints(480, 640, 6000, seed=3)
    ref = oracle.hashsift_responses(img, kps)
    fx = oracle.hashsift_responses_fixedpoint(img, kps)
    d = np.abs(fx - ref)
    assert d.max() <= 1.0
    assert (d > 0).mean() <= 1e-4
    _, b0 = oracle.hashsift_project(ref, 512)
    _, b1 = oracle.hashsift_project(fx, 512)
    assert np.count_nonzero(b0 != b1) <= int(1e-4 * b0.size)
    # the pinned probe vector: the model may move single elements, never more than one unit
    pimg, pk = probe_input()
    assert np.abs(oracle.hashsift_responses_fixedpoint(pimg, pk) - oracle.hashsift_responses(pimg, pk)).max() <= 1.0
