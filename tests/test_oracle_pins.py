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
