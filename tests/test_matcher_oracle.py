"""The matcher oracle (numpy restatement of cv::BFMatcher NORM_HAMMING semantics) against a naive loop."""
import numpy as np

from oracle import matcher_oracle as MO


def test_knn2_and_crosscheck_small():
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, size=(40, 32), dtype=np.uint8)
    t = rng.integers(0, 256, size=(55, 32), dtype=np.uint8)
    t[7] = t[3]; t[3] = q[5]; t[7] = q[5]            # a tie: rows 3 and 7 are both exact copies of query 5
    idx, dist = MO.knn2(q, t)
    for i in range(40):
        d = [bin(int.from_bytes(bytes(np.bitwise_xor(q[i], t[j])), "little")).count("1") for j in range(55)]
        order = sorted(range(55), key=lambda j: (d[j], j))[:2]
        assert list(idx[i]) == order and list(dist[i]) == [d[order[0]], d[order[1]]]
    assert list(idx[5]) == [3, 7] and list(dist[5]) == [0, 0]
    m, dd = MO.crosscheck(q, t)
    assert m[5] == 3 and dd[5] == 0
    # cross-check is symmetric: every kept pair is mutual
    m2, _ = MO.crosscheck(t, q)
    for i, j in enumerate(m):
        if j >= 0:
            assert m2[j] == i


def test_degenerate_sizes():
    q = np.zeros((3, 64), np.uint8)
    idx, dist = MO.knn2(q, np.zeros((1, 64), np.uint8))
    assert (idx[:, 0] == 0).all() and (idx[:, 1] == -1).all() and (dist[:, 1] == -1).all()
    m, d = MO.crosscheck(np.zeros((0, 32), np.uint8), np.zeros((4, 32), np.uint8))
    assert m.shape == (0,)
