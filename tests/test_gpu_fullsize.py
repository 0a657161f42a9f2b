"""Full-size parity for the BASELINE.json configurations C2-C5, HIP path through the C ABI against
the oracle on the same frames bench.py / tools/bench_configs.py time.  The reference's own test is full-image too
(tests/descriptor_test.cpp:19-75).  The oracle runs multi-threaded here (identical output:
tests/test_oracle_detector.py::test_thread_count_does_not_change_results); an 8K frame costs ~1 s on the GPU box's host."""
import os

import numpy as np
import pytest

from tools import workloads

pytestmark = pytest.mark.gpu

HS_T_ABS_TOL = 2e-3      # same tolerances as tests/test_gpu_parity.py::test_hashsift_compute_tolerance
HS_VEC_FRAC = 1e-4
HS_VEC_MAX = 4.0


@pytest.fixture(scope="module")
def cef():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cef_loader
    return cef_loader.load()


@pytest.fixture(scope="module")
def threaded_oracle(oracle):
    prev = oracle.set_threads(max(1, min(os.cpu_count() or 1, 64)))
    yield oracle
    oracle.set_threads(prev)


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _same_keypoints(kps, n, ref):
    assert n == ref["n"]
    g, r = kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32)
    for row, name in enumerate(["location", "response", "angle", "octave", "size"]):
        assert np.array_equal(g[row], r[row]), f"{name} row differs at {np.flatnonzero(g[row] != r[row])[:5]}"


def test_c2_4k_detect_only(cef, threaded_oracle):
    """C2: single 4K frame, detect-only (pyramid + FAST-9 + Harris + radius NMS + quota + IC angle), reference defaults."""
    import torch
    img = workloads.frame_c2()
    det = cef.EfficientFeatures.create(workloads.N40K, dtype=cef.EfficientFeatures.BAD_256)
    kps, cnt = det.detectAsync(_dev(img))
    torch.cuda.synchronize()
    ref = threaded_oracle.detect_and_compute(img, nfeatures=workloads.N40K, desc_type=-1)
    _same_keypoints(kps, int(cnt.item()), ref)
    st = det.lastLevelStats()
    for l in range(8):
        assert st[l]["n_candidates"] == ref["stats"]["n_candidates"][l]
        assert st[l]["n_after_nms"] == ref["stats"]["n_after_nms"][l]


@pytest.fixture(scope="module")
def c34(cef, threaded_oracle):
    """The C3 / C4 workload: 4K frame + EXACTLY 40 000 detector keypoints (tools/workloads.py explains the NMS radius)."""
    import torch
    img = workloads.frame_c34()
    d_img = _dev(img)
    det = cef.EfficientFeatures.create(workloads.N40K, 1.2, 8, 0, 20, workloads.C34_NMS_RADIUS, cef.EfficientFeatures.BAD_256)
    kps, cnt = det.detectAsync(d_img)
    torch.cuda.synchronize()
    n = int(cnt.item())
    ref = threaded_oracle.detect_and_compute(img, nfeatures=workloads.N40K, nonmax_radius=workloads.C34_NMS_RADIUS, desc_type=-1)
    _same_keypoints(kps, n, ref)
    assert n == workloads.N40K, "the compute-only configurations are quoted on 40 000 keypoints"
    k = cef.unpack_keypoints(kps[:, :n].cpu().numpy())
    kp4 = np.stack([k["x"].astype(np.float32), k["y"].astype(np.float32), np.full(n, 31, np.float32), k["angle"]], axis=1)
    return dict(img=img, d_img=d_img, kps=kps, n=n, kp4=kp4)


@pytest.mark.parametrize("nbits", [256, 512])
def test_c3_4k_40k_compute_bad(cef, threaded_oracle, c34, nbits):
    """C3: compute-only BAD256 / BAD512 on the 40 000 keypoints of a 4K frame (compute() semantics: the full-resolution
    image, size forced to 31, no blur; cuda_efficient_features.cpp:102-115, sample_benchmark.cpp:132-141).  Bit-exact."""
    import torch
    det = cef.EfficientFeatures.create(workloads.N40K, dtype=cef.EfficientFeatures.BAD_256 if nbits == 256 else cef.EfficientFeatures.BAD_512)
    desc = det.computeAsync(c34["d_img"], c34["kps"], n=c34["n"])
    torch.cuda.synchronize()
    want = threaded_oracle.bad_compute(c34["img"], c34["kp4"], nbits)
    got = desc.cpu().numpy()
    assert got.shape == (workloads.N40K, nbits // 8)
    assert np.array_equal(got, want), f"{np.count_nonzero((got != want).any(axis=1))} of 40000 descriptors differ"


@pytest.mark.parametrize("nbits", [256, 512])
def test_c4_4k_40k_compute_hashsift(cef, threaded_oracle, c34, nbits):
    """C4: compute-only HashSIFT256 / HashSIFT512 (MFMA projection) on the same 40 000 keypoints; the tolerances of
    test_hashsift_compute_tolerance, at full size.  (Round 6: 256 bits too -- VERDICT r5: the bench times both.)"""
    import torch
    det = cef.EfficientFeatures.create(workloads.N40K, dtype=cef.EfficientFeatures.HASH_SIFT_512 if nbits == 512 else cef.EfficientFeatures.HASH_SIFT_256)
    desc = det.computeAsync(c34["d_img"], c34["kps"], n=c34["n"]).cpu().numpy()
    hs = cef.HashSIFT.create(1.0, cef.HashSIFT.SIZE_512_BITS if nbits == 512 else cef.HashSIFT.SIZE_256_BITS)
    resp, T = hs.debug(c34["d_img"], _dev(c34["kp4"]), max_size=31.0)
    torch.cuda.synchronize()
    resp, T = resp.cpu().numpy(), T.cpu().numpy()
    want_resp = threaded_oracle.hashsift_responses(c34["img"], c34["kp4"])
    want_T, want_desc = threaded_oracle.hashsift_project(want_resp, nbits)
    d = np.abs(resp - want_resp)
    elem_rate, byte_rate = float((d > 0).mean()), float(np.count_nonzero(desc != want_desc)) / desc.size
    # measured rates of the 15.17 fixed-point histogram against the CPU float sums (DESIGN.md section 3 records them;
    # `pytest -s` shows them): the bound asserted here is 3x tighter than the stated tolerance, so an accuracy
    # regression shows up before the reference's 1e-4 is reached
    print(f"\nC4 HashSIFT{nbits} vs CPU reference arithmetic: {elem_rate:.2e} of the 129-vector elements differ by one unit, "
          f"{byte_rate:.2e} of the descriptor bytes differ ({workloads.N40K} keypoints)")
    assert d.max() <= HS_VEC_MAX and elem_rate <= HS_VEC_FRAC
    assert elem_rate <= 3e-5 and byte_rate <= 4e-5
    same = (d == 0).all(axis=1)
    assert np.abs(T[same] - want_T[same]).max() <= HS_T_ABS_TOL
    bits = np.unpackbits(desc, axis=1).astype(bool)
    wbits = np.unpackbits(want_desc, axis=1).astype(bool)
    decided = np.abs(want_T) > HS_T_ABS_TOL
    assert np.array_equal(bits[same][decided[same]], wbits[same][decided[same]])
    assert np.count_nonzero(desc != want_desc) <= max(1, int(1e-4 * desc.size))     # descriptor_test.cpp:72
    assert np.array_equal(bits, T > 0)


@pytest.mark.parametrize("k", [0, 1, 5])
def test_c5_8k_detect_and_compute_bad512(cef, threaded_oracle, k):
    """C5: 8K frames of the headline batch (seeds 1000 + k), detectAndCompute BAD512, 40 000 keypoints: every keypoint
    row and every descriptor byte equal the oracle's."""
    import torch
    img = workloads.frame_c5(k)
    det = cef.EfficientFeatures.create(workloads.N40K, dtype=cef.EfficientFeatures.BAD_512)
    kps, desc, cnt = det.detectAndComputeAsync(_dev(img))
    torch.cuda.synchronize()
    n = int(cnt.item())
    ref = threaded_oracle.detect_and_compute(img, nfeatures=workloads.N40K, desc_type=threaded_oracle.BAD_512)
    _same_keypoints(kps, n, ref)
    assert n == workloads.N40K
    assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"])
    if k == 0 and "EFX_BLUR_FORK" not in os.environ:
        # a caller that waits for every frame: from the second such call on, a frame of this size has its level blur on the
        # context's side stream beside select / emit / angle (efx_api.cpp, detect_common) -- same result, call after call,
        # and again inline when the calls come back to back
        dimg = _dev(img)
        for rep in range(4):
            if rep != 3:
                kps.zero_(); desc.zero_(); cnt.zero_()
                torch.cuda.synchronize()
            det.detectAndComputeAsync(dimg, kps, desc, cnt)
            if rep == 2: continue                             # the fourth call finds the stream busy
            torch.cuda.synchronize()
            _same_keypoints(kps, int(cnt.item()), ref)
            assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"])


def test_c5_8k_detect_and_compute_bad256(cef, threaded_oracle):
    """A C5 frame with the 256-bit BAD describer (only BAD512 was checked at 8K)."""
    import torch
    img = workloads.frame_c5(3)
    det = cef.EfficientFeatures.create(workloads.N40K, dtype=cef.EfficientFeatures.BAD_256)
    kps, desc, cnt = det.detectAndComputeAsync(_dev(img))
    torch.cuda.synchronize()
    n = int(cnt.item())
    ref = threaded_oracle.detect_and_compute(img, nfeatures=workloads.N40K, desc_type=threaded_oracle.BAD_256)
    _same_keypoints(kps, n, ref)
    assert n == workloads.N40K
    assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"])


def test_c5_8k_batch_of_four_equals_oracle(cef, threaded_oracle):
    """C5 as a frame-batched launch: four 8K frames on one context in ONE launch of every kernel (round 6), each frame against
    the oracle."""
    import torch
    imgs = [workloads.frame_c5(k) for k in (0, 1, 4, 6)]
    d = [_dev(im) for im in imgs]
    det = cef.EfficientFeatures.create(workloads.N40K, dtype=cef.EfficientFeatures.BAD_512)
    st = torch.cuda.Stream()
    kps = [torch.zeros((5, workloads.N40K), dtype=torch.float32, device="cuda") for _ in d]
    desc = [torch.zeros((workloads.N40K, 64), dtype=torch.uint8, device="cuda") for _ in d]
    cnt = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in d]
    torch.cuda.synchronize()
    cef.Batch([det], [st], d, kps, desc, cnt, workloads.N40K).run()
    torch.cuda.synchronize()
    for i, im in enumerate(imgs):
        ref = threaded_oracle.detect_and_compute(im, nfeatures=workloads.N40K, desc_type=threaded_oracle.BAD_512)
        n = int(cnt[i].item())
        _same_keypoints(kps[i], n, ref)
        assert n == workloads.N40K and np.array_equal(desc[i][:n].cpu().numpy(), ref["desc"])


def test_c5_8k_detect_and_compute_hashsift512(cef, threaded_oracle):
    import torch
    img = workloads.frame_c5(2)
    det = cef.EfficientFeatures.create(workloads.N40K, dtype=cef.EfficientFeatures.HASH_SIFT_512)
    kps, desc, cnt = det.detectAndComputeAsync(_dev(img))
    torch.cuda.synchronize()
    n = int(cnt.item())
    ref = threaded_oracle.detect_and_compute(img, nfeatures=workloads.N40K, desc_type=threaded_oracle.HASH_SIFT_512)
    _same_keypoints(kps, n, ref)
    got = desc[:n].cpu().numpy()
    assert np.count_nonzero(got != ref["desc"]) <= max(1, int(1e-4 * got.size))       # descriptor_test.cpp:72


def test_natural_8k_frame_sparse_form_by_the_previous_frames_density(cef, threaded_oracle):
    """An 8K frame with the statistics of photographs (1/f^1.3: 8 .. 20 FAST corners per tile): a context's first call takes
    harris_kernel, the following ones -- by the corner density the previous frame left in host memory -- harris_packed_kernel;
    then a corner-rich frame, and the sparse one again.  Every call equals the oracle: the choice is a matter of speed alone."""
    import torch
    from tools import synth
    (f,) = synth.powerlaw_frames_tiled(4320, 7680, seed=1000, betas=(1.3,))
    dense = workloads.frame_c5(0)
    det = cef.EfficientFeatures.create(workloads.N40K, dtype=cef.EfficientFeatures.BAD_512)
    refs = {}
    for name, img in (("sparse", f), ("sparse", f), ("sparse", f), ("dense", dense), ("sparse", f), ("sparse", f)):
        if name not in refs:
            refs[name] = threaded_oracle.detect_and_compute(img, nfeatures=workloads.N40K, desc_type=threaded_oracle.BAD_512)
        kps, desc, cnt = det.detectAndComputeAsync(_dev(img))
        torch.cuda.synchronize()
        n = int(cnt.item())
        _same_keypoints(kps, n, refs[name])
        assert np.array_equal(desc[:n].cpu().numpy(), refs[name]["desc"])


@pytest.mark.parametrize("kind", ["powerlaw", "powerlaw_dense", "blurred_edges"])
def test_natural_statistics_4k(cef, threaded_oracle, kind):
    """4K frames with the statistics of photographs (the reference tests on 11 photographs,
    tests/descriptor_test.cpp:25-36): 1/f texture (beta 1.3, and beta 1.0 where 17 % of the pixels are FAST corners and the
    10 % cap cuts) and shapes behind Gaussian defocus of four widths.  detectAndCompute BAD512 bit-exact, HashSIFT512 on the
    same frame inside the stated byte tolerance."""
    import torch
    from tools import synth
    rows, cols = workloads.K4
    img = {"powerlaw": lambda: synth.powerlaw_frame(rows, cols, seed=2001),
           "powerlaw_dense": lambda: synth.powerlaw_frame(rows, cols, seed=2002, beta=1.0),
           "blurred_edges": lambda: synth.blurred_edges_frame(rows, cols, seed=2003)}[kind]()
    d_img = _dev(img)
    det = cef.EfficientFeatures.create(workloads.N40K, dtype=cef.EfficientFeatures.BAD_512)
    kps, desc, cnt = det.detectAndComputeAsync(d_img)
    torch.cuda.synchronize()
    if det.overflowEvents() or int(cnt.item()) == 0:           # a frame denser than the density-sized arenas is void once
        kps, desc, cnt = det.detectAndComputeAsync(d_img)
        torch.cuda.synchronize()
    n = int(cnt.item())
    ref = threaded_oracle.detect_and_compute(img, nfeatures=workloads.N40K, desc_type=threaded_oracle.BAD_512)
    _same_keypoints(kps, n, ref)
    assert n > 5000
    if kind == "powerlaw_dense":
        assert ref["stats"]["n_candidates"][0] > ref["stats"]["n_after_cap"][0]      # the cap is active on this frame
    assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"])
    st = det.lastLevelStats()
    for l in range(8):
        assert st[l]["n_candidates"] == ref["stats"]["n_candidates"][l] and st[l]["n_after_nms"] == ref["stats"]["n_after_nms"][l]
    hs = cef.EfficientFeatures.create(workloads.N40K, dtype=cef.EfficientFeatures.HASH_SIFT_512)
    kps2, desc2, cnt2 = hs.detectAndComputeAsync(d_img)
    torch.cuda.synchronize()
    if int(cnt2.item()) == 0:
        kps2, desc2, cnt2 = hs.detectAndComputeAsync(d_img)
        torch.cuda.synchronize()
    ref2 = threaded_oracle.detect_and_compute(img, nfeatures=workloads.N40K, desc_type=threaded_oracle.HASH_SIFT_512)
    _same_keypoints(kps2, int(cnt2.item()), ref2)
    got = desc2[:n].cpu().numpy()
    nbad = int(np.count_nonzero(got != ref2["desc"]))
    print(f"\n{kind}: {n} keypoints, FAST corners {sum(ref['stats']['n_candidates'])}, HashSIFT512 bytes differing {nbad} of {got.size} ({nbad / got.size:.2e})")
    assert nbad <= max(4, int(1e-4 * got.size))


@pytest.mark.parametrize("nbytes", [32, 64])
@pytest.mark.parametrize("kind", ["random", "ties"])
def test_matcher_40k_by_40k_all_three_kernels(cef, monkeypatch, nbytes, kind):
    """The quoted matcher workload at its FULL size: knnMatch(k = 2) of 40 000 x 40 000 descriptors of 256 and
    512 bit through the FP4 (MX) matrix-core kernel, the int8 matrix-core kernel and the popcount kernel -- at this size the
    matrix-core kernels cut the train set into occupancy-sized chunks and merge the per-chunk best-twos (knn2_merge_kernel), a
    chunk count no smaller case reaches -- against oracle/matcher_oracle.c (OpenMP popcounts; sample_image_sequence.cpp:114-144).
    `ties`: descriptors drawn from a pool of 300 with noisy copies, so that most queries have several trains at their best and
    second-best distance and the lower train index must win across chunk boundaries."""
    import torch
    from oracle import matcher_oracle as MO
    n = 40000
    rng = np.random.default_rng(4000 + nbytes + (7 if kind == "ties" else 0))
    if kind == "random":
        q = rng.integers(0, 256, size=(n, nbytes), dtype=np.uint8)
        t = rng.integers(0, 256, size=(n, nbytes), dtype=np.uint8)
        sel = rng.permutation(n)[: n // 2]
        t[sel] = q[rng.permutation(n)[: n // 2]]                        # exact matches at random train positions
        near = rng.permutation(n)[: n // 8]
        t[near] = q[near] ^ (rng.random((len(near), nbytes)) < 0.05).astype(np.uint8) * np.uint8(4)
    else:
        pool = rng.integers(0, 256, size=(300, nbytes), dtype=np.uint8)
        q = pool[rng.integers(0, 300, n)] ^ ((rng.random((n, nbytes)) < 0.01).astype(np.uint8) * np.uint8(1))
        t = pool[rng.integers(0, 300, n)] ^ ((rng.random((n, nbytes)) < 0.01).astype(np.uint8) * np.uint8(1))
    q, t = np.ascontiguousarray(q), np.ascontiguousarray(t)
    widx, wdist = MO.knn2_c(q, t)
    if kind == "ties":
        assert (np.diff(wdist, axis=1) == 0).mean() > 0.2                  # many queries with best == second best
    for k in ("EFX_MATCH_NO_MFMA", "EFX_MATCH_NO_FP4"):
        monkeypatch.delenv(k, raising=False)
    dq, dt = _dev(q), _dev(t)
    for name, env in (("fp4", None), ("int8", "EFX_MATCH_NO_FP4"), ("popcount", "EFX_MATCH_NO_MFMA")):
        if env:
            monkeypatch.setenv(env, "1")
        m = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING)               # the knobs are read when a matcher is created
        if env:
            monkeypatch.delenv(env)
        idx, dist = m.knnMatch(dq, dt, 2)
        torch.cuda.synchronize()
        gd, gi = dist.cpu().numpy(), idx.cpu().numpy()
        bad = np.flatnonzero((gd != wdist).any(axis=1) | (gi != widx).any(axis=1))
        assert bad.size == 0, f"{name} kernel, {8 * nbytes} bit, {kind}: {bad.size} queries differ, first {bad[:5]}: got {gi[bad[:3]].tolist()} / {gd[bad[:3]].tolist()} want {widx[bad[:3]].tolist()} / {wdist[bad[:3]].tolist()}"
