"""The C oracle (oracle/efx_oracle.c) against a SECOND restatement of the reference's CPU descriptors (tests/second_opinion.py:
numpy, written from bad.cpp / hash_sift.cpp alone) -- VERDICT r4 item 3.  Neither is the reference (OpenCV is not in this image, so
parity stays "unpinned" beyond the four recorded hashes of tests/test_oracle_pins.py); what this module removes is the risk
that ONE transcriber misread a convention: every case below is a convention the recorded hashes do not reach -- angle == -1,
angle < 0, sizes 5 .. 110, scale factors 0.75 .. 2.5, keypoints on and outside the border, the clamped boxes of large keypoints,
saturate_cast's rounding.  BAD bytes must be equal bit for bit; HashSIFT patches and 129-vectors must be equal, descriptor bits
wherever the projection is not within float rounding of zero."""
import numpy as np
import pytest

from tests import second_opinion as so
from tools import synth


def _keypoints(rng, rows, cols, n, sizes, interior_only=False, margin_factor=1.0):
    k = np.zeros((n, 4), np.float32)
    k[:, 2] = rng.choice(np.asarray(sizes, np.float32), size=n)
    if interior_only:
        m = 0.9 * margin_factor * k[:, 2] + 8
        k[:, 0] = m + rng.random(n) * np.maximum(cols - 1 - 2 * m, 1)
        k[:, 1] = m + rng.random(n) * np.maximum(rows - 1 - 2 * m, 1)
    else:
        k[:, 0] = rng.uniform(-20, cols + 20, n)
        k[:, 1] = rng.uniform(-20, rows + 20, n)
    k[: n // 2, 0:2] = np.floor(k[: n // 2, 0:2])                      # detector keypoints have integer coordinates
    k[:, 3] = rng.uniform(0, 360, n)
    k[0::7, 3] = -1.0                                                  # the axis-aligned branch (bad.cpp:127-135)
    k[1::7, 3] = rng.uniform(-300, -0.01, len(k[1::7]))                # angle < 0 and != -1: cos = 1, sin = 0 (bad.cpp:138-139)
    k[2::7, 3] = 0.0
    k[3::7, 3] = np.float32(359.99)
    return k


@pytest.mark.parametrize("nbits", [256, 512])
@pytest.mark.parametrize("scale", [0.75, 1.0, 1.7, 2.5])
def test_bad_bytes_equal_second_restatement(oracle, nbits, scale):
    rng = np.random.default_rng(1000 * nbits + int(100 * scale))
    img = synth.powerlaw_frame(300, 420, seed=17, beta=1.1, contrast=60.0)
    sizes = [9, 13, 24.5, 31, 48, 77, 110]
    # interior keypoints (the integer path) of every size that fits the frame at this scale, and keypoints anywhere --
    # on the border, outside the frame -- for the clamped float path
    fit = [s for s in sizes if 2 * (0.9 * scale * s + 8) < 290]
    kin = _keypoints(rng, 300, 420, 160, fit, interior_only=True, margin_factor=scale)
    kany = _keypoints(rng, 300, 420, 240, [5, 7] + sizes)
    kany = kany[~((kany[:, 2] < 9) & (kany[:, 0] > 3) & (kany[:, 0] < 416) & (kany[:, 1] > 3) & (kany[:, 1] < 296))]     # tiny keypoints only where the border path takes them
    for name, kps in (("interior", kin), ("anywhere", kany)):
        want = so.bad_describe(img, kps, nbits, scale)
        got = oracle.bad_compute(img, kps, nbits, scale_factor=scale)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, f"{name}: {bad.size} of {len(kps)} descriptors differ, first keypoints {kps[bad[:4]].tolist()}"
    # both paths were really taken
    fw, fh = 420, 300
    bw = 32 * (scale * kany[:, 2] / 64) * 1.75
    on_border = (kany[:, 0] < bw) | (kany[:, 0] + bw >= fw) | (kany[:, 1] < bw) | (kany[:, 1] + bw >= fh)
    assert on_border.sum() > 50 and len(kin) > 100


def test_bad_interior_and_border_paths_agree_with_brute_force_means():
    """The two code paths of computeBAD are the same measurement (box mean difference against the threshold) wherever no box is
    clamped: the restatement's integer path equals a brute-force evaluation with pixel sums -- a check of the restatement
    itself (integral image layout, box extents, bit order), independent of either oracle."""
    rng = np.random.default_rng(5)
    img = synth.noise_frame(120, 160, seed=9)
    kps = _keypoints(rng, 120, 160, 24, [31], interior_only=True)
    boxes, thr = so.bad_params(256)
    got = so.bad_describe(img, kps, 256, 1.0)
    for n, (x, y, size, angle) in enumerate(kps):
        s = np.float32(size) / np.float32(32)
        if angle == -1 or angle < 0:
            c, sn = np.float32(1), np.float32(0)
        else:
            c, sn = np.float32(np.cos(float(angle) * 0.017453292519943295)), np.float32(np.sin(float(angle) * 0.017453292519943295))
        bits = []
        for (x1, x2, y1, y2, r), t in zip(boxes, thr):
            def centre(bx, by):
                if angle == -1:
                    return int(np.float32(s * bx + (np.float32(-0.5) * s * 32 + x)) + np.float32(0.5)), int(np.float32(s * by + (-s * np.float32(0.5) * 32 + y)) + np.float32(0.5))
                m02 = (-s * c + s * sn) * 32 * np.float32(0.5) + x
                m12 = (-s * sn - s * c) * 32 * np.float32(0.5) + y
                return int(np.float32(s * c * bx + -s * sn * by + m02) + np.float32(0.5)), int(np.float32(s * sn * bx + s * c * by + m12) + np.float32(0.5))
            rr = int(s * r + np.float32(0.5))
            (ax, ay), (bx_, by_) = centre(x1, y1), centre(x2, y2)
            sa = int(img[ay - rr:ay + rr + 1, ax - rr:ax + rr + 1].astype(np.int64).sum())
            sb = int(img[by_ - rr:by_ + rr + 1, bx_ - rr:bx_ + rr + 1].astype(np.int64).sum())
            side = 2 * rr + 1
            bits.append(np.float32(sa - sb) <= np.float32(t * np.float32(side * side)))
        assert np.array_equal(np.packbits(np.array(bits, np.uint8)), got[n]), f"keypoint {n}"


@pytest.mark.parametrize("crop", [0.75, 1.0, 2.0])
def test_hashsift_patch_vector_and_bits_equal_second_restatement(oracle, crop):
    rng = np.random.default_rng(int(crop * 100))
    img = synth.blurred_edges_frame(260, 340, seed=23)
    kps = _keypoints(rng, 260, 340, 120, [5, 12, 31, 31, 47.5, 90])
    # patches (rectifyPatch + warpAffineLinear): pixels outside the frame are 0, angle < 0 means no rotation
    for i in range(0, len(kps), 5):
        assert np.array_equal(oracle.hashsift_patch(img, kps[i], crop), so.hashsift_patch(img, kps[i], crop)), f"patch of keypoint {kps[i].tolist()}"
    want, pre = so.hashsift_vectors(img, kps, crop)
    got = oracle.hashsift_responses(img, kps, crop)
    assert got.shape == want.shape == (len(kps), 129)
    diff = np.nonzero(got != want)
    assert diff[0].size == 0, f"{diff[0].size} of {got.size} vector elements differ, first (keypoint, element) {list(zip(diff[0][:4].tolist(), diff[1][:4].tolist()))}"
    # saturate_cast<uchar>(512 v): cvRound's round-half-to-even, then the clamp -- both restatements saw the same rule on every
    # element, including the ones closest to a tie
    frac = np.abs(pre - np.floor(pre) - 0.5)
    assert (want[:, 1:] <= 255).all() and (want[:, 1:] >= 0).all() and frac.min() < 5e-3
    for nbits in (256, 512):
        bytes_want, T = so.hashsift_bits(want, nbits)
        T_got, bytes_got = oracle.hashsift_project(got, nbits)
        assert np.abs(T_got.astype(np.float64) - T).max() < 1e-3     # same projection (double accumulation, any summation order)
        decided = np.abs(T) > 1e-3                                   # ... which cannot move these across zero
        bg = np.unpackbits(bytes_got, axis=1).astype(bool)
        bw = np.unpackbits(bytes_want, axis=1).astype(bool)
        assert np.array_equal(bg[decided], bw[decided])
        assert decided.mean() > 0.999


def test_saturate_cast_rule_on_exact_ties():
    """cv::saturate_cast<uchar>(float) rounds halves to even (cvRound) and clamps: the restatement's rule on exact ties, and the
    oracle's through the one place it applies it -- a 129-vector element is an integer in [0, 255]."""
    v = np.array([0.5, 1.5, 2.5, 254.5, 255.5, 300.0, -0.5, -3.0], np.float32)
    assert np.clip(np.rint(v), 0, 255).tolist() == [0, 2, 2, 254, 255, 255, 0, 0]
