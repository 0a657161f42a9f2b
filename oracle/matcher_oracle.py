"""CPU ORACLE of cv::BFMatcher(NORM_HAMMING) as the reference's samples use it (test infrastructure only).
OpenCV is a third-party dependency that is absent here (unpinned version >= 4.6, README.md:80-81); the published
algorithm is restated: Hamming distance = popcount(xor); knnMatch keeps the k smallest distances per query, scanning
the train descriptors in order with a strict `<` (so ties go to the lower train index); crossCheck keeps (i, j)
only if j is i's nearest train and i is j's nearest query.  Call sites: samples/sample_feature_matching.cpp:99-101,
samples/sample_image_sequence.cpp:81,114-144.  parity unpinned (no OpenCV, no golden vectors)."""
import numpy as np

_POP = np.array([bin(i).count("1") for i in range(256)], dtype=np.int32)


def hamming_matrix(query, train):
    q = np.ascontiguousarray(query, dtype=np.uint8)
    t = np.ascontiguousarray(train, dtype=np.uint8)
    out = np.zeros((q.shape[0], t.shape[0]), dtype=np.int32)
    for i in range(q.shape[0]):
        out[i] = _POP[np.bitwise_xor(q[i][None, :], t)].sum(axis=1)
    return out


def knn2(query, train):
    d = hamming_matrix(query, train)
    nq, nt = d.shape
    idx = np.full((nq, 2), -1, np.int32)
    dist = np.full((nq, 2), -1, np.int32)
    for i in range(nq):
        order = np.argsort(d[i], kind="stable")[:2]          # stable: ties -> lower train index
        idx[i, :len(order)] = order
        dist[i, :len(order)] = d[i][order]
    return idx, dist


def crosscheck(query, train):
    d = hamming_matrix(query, train)
    nq, nt = d.shape
    m = np.full(nq, -1, np.int32)
    dd = np.full(nq, -1, np.int32)
    if nq == 0 or nt == 0:
        return m, dd
    q2t = d.argmin(axis=1)            # first minimum = lower index on ties
    t2q = d.argmin(axis=0)
    for i in range(nq):
        dd[i] = d[i, q2t[i]]
        if t2q[q2t[i]] == i:
            m[i] = q2t[i]
    return m, dd


def knn2_c(query, train):
    """knn2 at full size: oracle/matcher_oracle.c (OpenMP, 64-bit popcounts), same tie rule; desc bytes a multiple of 8."""
    import ctypes
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "libefx_matcher_oracle.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", here, "libefx_matcher_oracle.so"])
    lib = ctypes.CDLL(so)
    q = np.ascontiguousarray(query, dtype=np.uint8)
    t = np.ascontiguousarray(train, dtype=np.uint8)
    assert q.shape[1] == t.shape[1] and q.shape[1] % 8 == 0 and q.shape[1] <= 128
    idx = np.full((q.shape[0], 2), -1, np.int32)
    dist = np.full((q.shape[0], 2), -1, np.int32)
    lib.efxo_knn2_hamming(q.ctypes.data_as(ctypes.c_void_p), q.shape[0], t.ctypes.data_as(ctypes.c_void_p), t.shape[0], q.shape[1],
                          idx.ctypes.data_as(ctypes.c_void_p), dist.ctypes.data_as(ctypes.c_void_p))
    return idx, dist
