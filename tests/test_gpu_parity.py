"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI, against the
CPU oracle on the same seeded inputs.  Integer / byte / index results must be bit-exact; HashSIFT has a stated
float tolerance before thresholding (reference tests: tests/descriptor_test.cpp:19-75, which allow 2e-5 / 1e-4
differing bytes GPU-vs-CPU; here BAD must be exact)."""
import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu

import os


@pytest.fixture(scope="module")
def cef():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cef_loader
    return cef_loader.load()


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    return torch


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _detect_both(cef, torch, oracle, img, desc_type=-1, **kw):
    nfeatures = kw.get("nfeatures", 5000)
    det = cef.EfficientFeatures.create(nfeatures, kw.get("scale_factor", 1.2), kw.get("nlevels", 8),
                                       kw.get("first_level", 0), kw.get("fast_threshold", 20),
                                       kw.get("nonmax_radius", 15), max(desc_type, 0))
    d_img = _dev(torch, img)
    if desc_type >= 0:
        kps, desc, cnt = det.detectAndComputeAsync(d_img)
    else:
        kps, cnt = det.detectAsync(d_img)
        desc = None
    torch.cuda.synchronize()
    n = int(cnt.item())
    assert det.lastCount() == n
    ref = oracle.detect_and_compute(img, desc_type=desc_type, **kw)
    got = dict(n=n, kps=kps[:, :n].cpu().numpy(), desc=None if desc is None else desc[:n].cpu().numpy(),
               stats=det.lastLevelStats(), det=det)
    return got, ref


def _assert_same_keypoints(got, ref):
    st = got["stats"]
    for l, s in enumerate(st):
        assert s["n_candidates"] == ref["stats"]["n_candidates"][l], f"level {l}: FAST corner count"
        assert s["n_after_nms"] == ref["stats"]["n_after_nms"][l], f"level {l}: NMS survivor count"
        assert s["n_kept"] == ref["stats"]["n_kept"][l], f"level {l}: kept count"
    assert got["n"] == ref["n"]
    g, r = got["kps"].view(np.uint32), ref["kps"].view(np.uint32)
    for row, name in enumerate(["location", "response", "angle", "octave", "size"]):
        bad = np.nonzero(g[row] != r[row])[0]
        assert bad.size == 0, f"{name} row differs at {bad[:8]} (of {bad.size}); got {got['kps'][row][bad[:4]]} want {ref['kps'][row][bad[:4]]}"


@pytest.mark.parametrize("shape", [(480, 640), (501, 703), (720, 1280), (1080, 1920), (1520, 2704), (2160, 3840)])
def test_pyramid_levels_bit_exact(cef, torch_mod, oracle, shape, monkeypatch):
    """Spec S5 resize chain (cuda_efficient_features.cpp:136-157).  The shapes cover every tile edge the tower launch
    picks (plan_tower: 12, 20, 28, 36 with one workgroup per CU, 36 with two; a 4K frame takes the row-walking chain by default
    since round 5: EFX_TOWER_MAX_PX, read when a context is created, sends it through the tower here)."""
    monkeypatch.setenv("EFX_TOWER_MAX_PX", "9000000")
    img = synth.synth_frame(shape[0], shape[1], seed=11)
    det = cef.EfficientFeatures.create(1000)
    d_img = _dev(torch_mod, img)            # level 0 aliases the caller's image: keep it alive
    det.detectAsync(d_img)
    torch_mod.cuda.synchronize()
    for level in range(8):
        got = det.copyLevel(level, shape[0], shape[1]).cpu().numpy()
        want = oracle.pyramid_level(img, level)
        assert got.shape == want.shape
        assert np.array_equal(got, want), f"level {level}: {np.count_nonzero(got != want)} pixels differ"


@pytest.mark.parametrize("shape,nfeatures", [((480, 640), 1000), ((501, 703), 5000), ((1080, 1920), 40000), ((1080, 1920), 2000)])
def test_detect_bit_exact(cef, torch_mod, oracle, shape, nfeatures):
    """detectAsync == oracle: same keypoints, same canonical order, all five rows bit for bit."""
    img = synth.synth_frame(shape[0], shape[1], seed=1000)
    got, ref = _detect_both(cef, torch_mod, oracle, img, nfeatures=nfeatures)
    _assert_same_keypoints(got, ref)


def test_detect_unaligned_pitch(cef, torch_mod, oracle):
    """A view with an odd row pitch takes the byte-load path of the tile loader."""
    img = synth.synth_frame(400, 601, seed=5)
    big = np.zeros((400, 777), np.uint8)
    big[:, 3:604] = img
    d = _dev(torch_mod, big)[:, 3:604]
    det = cef.EfficientFeatures.create(3000)
    kps, cnt = det.detectAsync(d)
    torch_mod.cuda.synchronize()
    n = int(cnt.item())
    ref = oracle.detect_and_compute(img, nfeatures=3000)
    assert n == ref["n"]
    assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))


@pytest.mark.parametrize("kw", [dict(first_level=2), dict(nlevels=4), dict(nonmax_radius=0), dict(nonmax_radius=5),
                                dict(nonmax_radius=16), dict(nonmax_radius=33), dict(fast_threshold=5),
                                dict(fast_threshold=60), dict(scale_factor=1.5, nlevels=5), dict(nlevels=1)])
def test_detect_parameters(cef, torch_mod, oracle, kw):
    img = synth.synth_frame(600, 800, seed=21)
    got, ref = _detect_both(cef, torch_mod, oracle, img, nfeatures=3000, **kw)
    _assert_same_keypoints(got, ref)


def test_candidate_cap_on_noise(cef, torch_mod, oracle):
    """> 10 % FAST corners: the cap is applied in canonical order (spec S2, cuda_fast.cu:245)."""
    img = synth.noise_frame(300, 400, seed=7)
    got, ref = _detect_both(cef, torch_mod, oracle, img, nfeatures=5000)
    assert any(c > k for c, k in zip(ref["stats"]["n_candidates"], ref["stats"]["n_after_cap"])), "cap not exercised"
    _assert_same_keypoints(got, ref)


def test_tiny_and_empty(cef, torch_mod, oracle):
    flat = np.full((200, 300), 77, np.uint8)
    got, ref = _detect_both(cef, torch_mod, oracle, flat, nfeatures=500)
    assert got["n"] == 0 and ref["n"] == 0
    small = synth.synth_frame(40, 50, seed=3)          # upper levels have no pixel inside the 15-px border
    got, ref = _detect_both(cef, torch_mod, oracle, small, nfeatures=500)
    _assert_same_keypoints(got, ref)


def test_capacity_smaller_than_n(cef, torch_mod, oracle):
    img = synth.synth_frame(480, 640, seed=1000)
    det = cef.EfficientFeatures.create(5000)
    kps, cnt = det.detectAsync(_dev(torch_mod, img), capacity=100)
    torch_mod.cuda.synchronize()
    assert int(cnt.item()) == 100
    ref = oracle.detect_and_compute(img, nfeatures=5000, capacity=100)
    assert np.array_equal(kps[:, :100].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))


@pytest.mark.parametrize("desc_type", [0, 1])
@pytest.mark.parametrize("shape", [(480, 640), (1080, 1920)])
def test_detect_and_compute_bad_bit_exact(cef, torch_mod, oracle, desc_type, shape):
    """detectAndCompute: blurred level (spec S6) + BAD at level coordinates (cuda_efficient_features.cpp:302-307)."""
    img = synth.synth_frame(shape[0], shape[1], seed=1001)
    got, ref = _detect_both(cef, torch_mod, oracle, img, desc_type=desc_type, nfeatures=8000)
    _assert_same_keypoints(got, ref)
    bad = np.nonzero((got["desc"] != ref["desc"]).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of {got['n']} descriptors differ, first rows {bad[:8]}"


@pytest.mark.parametrize("nbits", [256, 512])
def test_compute_bad_random_keypoints(cef, torch_mod, oracle, nbits):
    """compute(): bit-exact BAD bytes for identical (image, x, y, size, angle), including border keypoints,
    angle == -1 and angle < 0 (bad.cpp:127,138; tests/descriptor_test.cpp:19-46)."""
    img = synth.synth_frame(480, 640, seed=4)
    kps = synth.random_keypoints(480, 640, 3000, seed=9)
    enum = cef.BAD.SIZE_256_BITS if nbits == 256 else cef.BAD.SIZE_512_BITS
    bad = cef.BAD.create(1.0, enum)
    got = bad.compute(img, kps)
    want = oracle.bad_compute(img, kps, nbits)
    diff = np.nonzero((got != want).any(axis=1))[0]
    assert diff.size == 0, f"{diff.size} descriptors differ, first {diff[:8]}, kps {kps[diff[:4]]}"


def test_compute_bad_matches_reference_hash(cef):
    """The SURVEY Appendix B known-answer vector of the reference CPU code, through the HIP path."""
    from tests.lcg_probe import REFERENCE_HASHES, fnv1a32, probe_input
    img, kps = probe_input()
    for nbits, enum in ((256, cef.BAD.SIZE_256_BITS), (512, cef.BAD.SIZE_512_BITS)):
        got = cef.BAD.create(1.0, enum).compute(img, kps)
        assert fnv1a32(got) == REFERENCE_HASHES[("bad", nbits)]


@pytest.mark.parametrize("size,scale", [(31.0, 1.0), (12.0, 1.0), (64.0, 1.0), (31.0, 2.5), (5.0, 5.0)])
def test_compute_bad_sizes(cef, oracle, size, scale):
    img = synth.synth_frame(480, 640, seed=6)
    kps = synth.random_keypoints(480, 640, 800, seed=13, size=size)
    got = cef.BAD.create(scale, cef.BAD.SIZE_256_BITS).compute(img, kps)
    want = oracle.bad_compute(img, kps, 256, scale_factor=scale)
    assert np.array_equal(got, want)


def test_compute_async_5xn_forces_size_31(cef, torch_mod, oracle):
    """computeAsync on the detector's 5xN matrix: size forced to 31, raw image, no blur
    (cuda_efficient_features.cpp:102-115, .cu:250-263; sample_benchmark.cpp:132-141)."""
    img = synth.synth_frame(480, 640, seed=1000)
    det = cef.EfficientFeatures.create(3000, dtype=cef.EfficientFeatures.BAD_512)
    d_img = _dev(torch_mod, img)
    kps, cnt = det.detectAsync(d_img)
    torch_mod.cuda.synchronize()
    n = int(cnt.item())
    desc = det.computeAsync(d_img, kps, n=n)
    torch_mod.cuda.synchronize()
    k = cef.unpack_keypoints(kps[:, :n].cpu().numpy())
    kp4 = np.stack([k["x"].astype(np.float32), k["y"].astype(np.float32), np.full(n, 31, np.float32), k["angle"]], axis=1)
    want = oracle.bad_compute(img, kp4, 512)
    assert np.array_equal(desc.cpu().numpy(), want)


def _matrix_5xn(torch_mod, rows, cols, n, seed):
    """A 5 x n keypoint matrix as detectAsync writes it: integer positions anywhere in the frame (corners and borders
    included), every angle convention; SIZE / OCTAVE rows hold values computeAsync must ignore (.cu:250-263)."""
    k = synth.random_keypoints(rows, cols, n, seed=seed)
    x = np.floor(k[:, 0]).astype(np.int64).clip(0, cols - 1); y = np.floor(k[:, 1]).astype(np.int64).clip(0, rows - 1)
    x[:4] = [0, cols - 1, 0, cols - 1]; y[:4] = [0, 0, rows - 1, rows - 1]
    m = np.zeros((5, n), dtype=np.float32)
    m[0] = ((x & 0xffff) | ((y & 0xffff) << 16)).astype(np.uint32).view(np.float32)
    m[1] = 1.0
    m[2] = k[:, 3]
    m[3] = np.arange(n, dtype=np.int32).view(np.float32) % 7
    m[4] = 77.0
    kp4 = np.stack([x.astype(np.float32), y.astype(np.float32), np.full(n, 31, np.float32), k[:, 3]], axis=1)
    return torch_mod.from_numpy(m).cuda(), kp4


@pytest.mark.parametrize("nbits", [256, 512])
@pytest.mark.parametrize("rows,cols,offset,pitch", [(480, 640, 0, 640), (203, 301, 0, 304), (203, 301, 1, 303), (40, 61, 0, 64),
                                                    (47, 200, 0, 200), (300, 45, 2, 46)])
def test_compute_async_wave_per_keypoint_kernel(cef, torch_mod, oracle, monkeypatch, nbits, rows, cols, offset, pitch):
    """Config C3's kernel (bad_raw_kernel: computeAsync on detector-sized keypoints, a wave per keypoint): equals the
    oracle bit for bit on frames larger and SMALLER than the 48 x 48 window, on unaligned bases / pitches (byte path), for
    border and corner keypoints; and equals the generic one-workgroup-per-keypoint kernel (EFX_BAD_NO_RAW)."""
    rng = np.random.default_rng(rows * 1000 + cols)
    img = synth.synth_frame(rows, cols, seed=rows + cols) if min(rows, cols) >= 64 else rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    buf = torch_mod.zeros(rows * pitch + 16, dtype=torch_mod.uint8, device="cuda")
    d_img = buf[offset:offset + rows * pitch].view(rows, pitch)[:, :cols]
    d_img.copy_(torch_mod.from_numpy(img).cuda())
    n = 1500
    d_kps, kp4 = _matrix_5xn(torch_mod, rows, cols, n, seed=nbits + rows)
    dtype = cef.EfficientFeatures.BAD_256 if nbits == 256 else cef.EfficientFeatures.BAD_512
    monkeypatch.delenv("EFX_BAD_NO_RAW", raising=False)
    det = cef.EfficientFeatures.create(n, dtype=dtype)
    monkeypatch.setenv("EFX_BAD_NO_RAW", "1")
    det_generic = cef.EfficientFeatures.create(n, dtype=dtype)
    monkeypatch.delenv("EFX_BAD_NO_RAW")
    desc = det.computeAsync(d_img, d_kps)
    desc2 = det_generic.computeAsync(d_img, d_kps)
    torch_mod.cuda.synchronize()
    want = oracle.bad_compute(img, kp4, nbits)
    assert np.array_equal(desc.cpu().numpy(), want)
    assert torch_mod.equal(desc, desc2)
    # unaligned descriptor rows (byte stores)
    out = torch_mod.zeros((n, nbits // 8 + 3), dtype=torch_mod.uint8, device="cuda")[:, 3:]
    det.computeAsync(d_img, d_kps, descriptors=out)
    torch_mod.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 67])
def test_compute_async_few_keypoints(cef, torch_mod, oracle, n):
    """The wave-per-keypoint kernel runs four keypoints per workgroup: counts that do not fill the last workgroup, and the
    device count (N) smaller than the matrix."""
    img = synth.synth_frame(300, 400, seed=8)
    d_img = _dev(torch_mod, img)
    d_kps, kp4 = _matrix_5xn(torch_mod, 300, 400, 80, seed=n)
    det = cef.EfficientFeatures.create(80, dtype=cef.EfficientFeatures.BAD_512)
    desc = det.computeAsync(d_img, d_kps, n=n)
    torch_mod.cuda.synchronize()
    assert np.array_equal(desc.cpu().numpy(), oracle.bad_compute(img, kp4[:n], 512))


# ---- HashSIFT: float tolerance before thresholding ----
HS_T_ABS_TOL = 2e-3      # |T_hip - T_oracle| for identical 129-vectors: fp32 FMA chain (MFMA) vs double accumulation
                         # of 129 products; |T| is typically ~15, at most ~400; measured max 2.2e-4
HS_VEC_FRAC = 1e-4       # fraction of 129-vector elements allowed to differ by one unit: the HIP histogram is summed in
                         # 15.17 fixed point (order independent), the CPU loop in sequentially rounded floats -- measured
                         # 1.4e-5 (2e-5 of the descriptor bytes; the reference's own GPU-vs-CPU bound is 1e-4); cosf/sinf of the keypoint angle are glibc's (csrc/glibc_sincosf.h), expf / atan2f host tables
HS_VEC_MAX = 4.0         # a one-grey-level flip of a patch pixel moves an element by a few units


@pytest.mark.parametrize("nbits", [256, 512])
def test_hashsift_compute_tolerance(cef, torch_mod, oracle, nbits):
    img = synth.synth_frame(480, 640, seed=4)
    kps = synth.random_keypoints(480, 640, 2000, seed=17)
    enum = cef.HashSIFT.SIZE_256_BITS if nbits == 256 else cef.HashSIFT.SIZE_512_BITS
    hs = cef.HashSIFT.create(1.0, enum)
    d_img = _dev(torch_mod, img)
    d_k = _dev(torch_mod, kps)
    resp, T = hs.debug(d_img, d_k, max_size=31.0)
    desc = hs.computeAsync(d_img, d_k, max_size=31.0)
    torch_mod.cuda.synchronize()
    resp, T, desc = resp.cpu().numpy(), T.cpu().numpy(), desc.cpu().numpy()
    want_resp = oracle.hashsift_responses(img, kps)
    want_T, want_desc = oracle.hashsift_project(want_resp, nbits)
    d = np.abs(resp - want_resp)
    assert d.max() <= HS_VEC_MAX, f"129-vector element off by {d.max()}"
    assert (d > 0).mean() <= HS_VEC_FRAC, f"{(d > 0).mean():.2e} of the 129-vector elements differ"
    # projection of identical vectors: fp32 FMA chain (MFMA) vs double accumulation
    same = (d == 0).all(axis=1)
    assert np.abs(T[same] - want_T[same]).max() <= HS_T_ABS_TOL
    # bits must agree wherever |T| exceeds the tolerance
    bits = np.unpackbits(desc, axis=1).astype(bool)
    wbits = np.unpackbits(want_desc, axis=1).astype(bool)
    decided = np.abs(want_T) > HS_T_ABS_TOL
    assert np.array_equal(bits[same][decided[same]], wbits[same][decided[same]])
    # and overall the reference's own tolerance (1e-4 of the bytes, descriptor_test.cpp:72) must hold
    assert np.count_nonzero(desc != want_desc) <= max(1, int(1e-4 * desc.size))
    # packed bits must be exactly sign(T) of the HIP path
    assert np.array_equal(bits, T > 0)


def test_hashsift_vectors_equal_fixed_point_model_bit_exact(cef, torch_mod, oracle):
    """The kernel against the CPU model of ITS OWN arithmetic (fixed-point histogram sums): every 129-vector element
    must match (cosf / sinf of the keypoint angle included: csrc/glibc_sincosf.h restates the host libm's)."""
    img = synth.synth_frame(480, 640, seed=4)
    kps = synth.random_keypoints(480, 640, 4000, seed=23)
    hs = cef.HashSIFT.create(1.0, cef.HashSIFT.SIZE_256_BITS)
    resp, _ = hs.debug(_dev(torch_mod, img), _dev(torch_mod, kps), max_size=31.0)
    torch_mod.cuda.synchronize()
    want = oracle.hashsift_responses_fixedpoint(img, kps)
    rows_off = np.nonzero((resp.cpu().numpy() != want).any(axis=1))[0]
    assert rows_off.size == 0, f"{rows_off.size} of 4000 vectors differ from the fixed-point model: {rows_off[:8]}"


@pytest.mark.parametrize("offset,pitch", [(0, 640), (3, 777), (1, 642)])
def test_hashsift_window_paths(cef, torch_mod, oracle, offset, pitch):
    """compute-only HashSIFT stages a keypoint's raw window in LDS as aligned dwords; images whose base or pitch is not
    4-byte aligned and keypoints whose window exceeds 120 px gather from memory instead.  All paths against the
    fixed-point model of the device arithmetic, bit for bit, with keypoint sizes from 3 to 150 in one call."""
    img = synth.synth_frame(400, 601, seed=5)
    big = np.zeros((400, max(pitch, 601 + offset)), np.uint8)
    big[:, offset:offset + 601] = img
    d = _dev(torch_mod, big)[:, offset:offset + 601]
    kps = synth.random_keypoints(400, 601, 1500, seed=29)
    kps[:, 2] = np.resize(np.array([3, 9.5, 31, 31, 31, 48, 64, 80, 100, 150], np.float32), kps.shape[0])
    hs = cef.HashSIFT.create(1.0, cef.HashSIFT.SIZE_256_BITS)
    resp, _ = hs.debug(d, _dev(torch_mod, kps), max_size=150.0)
    torch_mod.cuda.synchronize()
    want = oracle.hashsift_responses_fixedpoint(img, kps)
    rows_off = np.nonzero((resp.cpu().numpy() != want).any(axis=1))[0]
    assert rows_off.size == 0, f"{rows_off.size} of 1500 vectors differ from the fixed-point model: {rows_off[:8]}, sizes {kps[rows_off[:8], 2]}"


@pytest.mark.parametrize("desc_type", [2, 3])
def test_detect_and_compute_hashsift(cef, torch_mod, oracle, desc_type):
    img = synth.synth_frame(480, 640, seed=1002)
    got, ref = _detect_both(cef, torch_mod, oracle, img, desc_type=desc_type, nfeatures=4000)
    _assert_same_keypoints(got, ref)
    assert np.count_nonzero(got["desc"] != ref["desc"]) <= max(1, int(1e-4 * got["desc"].size))


def test_host_api_round_trip(cef, oracle):
    """detect / compute / detectAndCompute with host images (cv::Mat branch, cuda_efficient_features.cpp:197-213)."""
    img = synth.synth_frame(480, 640, seed=1000)
    det = cef.EfficientFeatures.create(3000, dtype=cef.EfficientFeatures.BAD_256)
    kps = det.detect(img)
    ref = oracle.detect_and_compute(img, nfeatures=3000)
    k = oracle.unpack_keypoints(ref["kps"])
    assert len(kps) == ref["n"]
    assert np.array_equal(kps["x"], k["x"].astype(np.float32)) and np.array_equal(kps["y"], k["y"].astype(np.float32))
    assert np.array_equal(kps["angle"], k["angle"]) and np.array_equal(kps["octave"], k["octave"])
    assert np.array_equal(kps["size"], k["size"]) and np.array_equal(kps["response"], k["response"])
    assert (kps["class_id"] == -1).all()
    # compute() honours kp.size (cuda_efficient_features.cpp:125): level-0 coordinates, scaled sizes, raw image
    desc = det.compute(img, kps)
    kp4 = np.stack([kps["x"], kps["y"], kps["size"], kps["angle"]], axis=1)
    assert np.array_equal(desc, oracle.bad_compute(img, kp4, 256))
    kps2, desc2 = det.detectAndCompute(img)
    ref2 = oracle.detect_and_compute(img, nfeatures=3000, desc_type=oracle.BAD_256)
    assert len(kps2) == ref2["n"] and np.array_equal(desc2, ref2["desc"])


def test_errors(cef, torch_mod):
    det = cef.EfficientFeatures.create(100)
    with pytest.raises(cef.EfxError):
        det.detect(np.zeros((10, 10, 3), np.uint8))           # not 8UC1
    with pytest.raises(cef.EfxError):
        det.detectAndCompute(np.zeros((64, 64), np.uint8), useProvidedKeypoints=True)
    with pytest.raises(cef.EfxError):
        det.setNLevels(0)
    with pytest.raises(cef.EfxError):
        cef.BAD.create(1.0, 123)                              # n_bits should be SIZE_512_BITS or SIZE_256_BITS
    with pytest.raises(cef.EfxError):
        det.computeAsync(torch_mod.zeros((64, 64), dtype=torch_mod.uint8, device="cuda"),
                         torch_mod.zeros((4, 10), dtype=torch_mod.float32, device="cuda"))
    # setters / getters round trip (cuda_efficient_features.cpp:355-379)
    det.setMaxFeatures(123); det.setScaleFactor(1.3); det.setNLevels(5); det.setFirstLevel(1)
    det.setFastThreshold(11); det.setNonmaxRadius(7); det.setDescriptorType(cef.EfficientFeatures.BAD_512)
    assert (det.getMaxFeatures(), det.getNLevels(), det.getFirstLevel(), det.getFastThreshold(), det.getNonmaxRadius(),
            det.getDescriptorType()) == (123, 5, 1, 11, 7, 1)
    assert abs(det.getScaleFactor() - 1.3) < 1e-6
    assert det.descriptorSize() == 64 and det.descriptorType() == 0 and det.defaultNorm() == 6


def test_determinism(cef, torch_mod):
    """Same frame twice on the same context and on a fresh one: identical bytes (the reference's atomic
    append order is nondeterministic; ours is canonical)."""
    img = synth.synth_frame(720, 1280, seed=1003)
    d_img = _dev(torch_mod, img)
    outs = []
    det = cef.EfficientFeatures.create(6000, dtype=cef.EfficientFeatures.BAD_256)
    for i in range(3):
        if i == 2:
            det = cef.EfficientFeatures.create(6000, dtype=cef.EfficientFeatures.BAD_256)
        kps, desc, cnt = det.detectAndComputeAsync(d_img)
        torch_mod.cuda.synchronize()
        n = int(cnt.item())
        outs.append((n, kps[:, :n].cpu().numpy().tobytes(), desc[:n].cpu().numpy().tobytes()))
    assert outs[0] == outs[1] == outs[2]


# ---- brute-force Hamming matcher ----
@pytest.mark.parametrize("nbytes", [32, 64])
@pytest.mark.parametrize("nq,nt", [(1000, 1300), (257, 3), (5, 1), (300, 300), (4097, 3001), (129, 65), (2500, 64)])
def test_matcher_knn2_and_crosscheck(cef, torch_mod, nbytes, nq, nt):
    """knnMatch(k=2) and crossCheck match are exact (integer distances, ties to the lower train index).  Sets of at least
    128 x 64 descriptors go through the int8 matrix-core kernel, smaller ones through the popcount kernel."""
    from oracle import matcher_oracle as MO
    rng = np.random.default_rng(nq * 7 + nt + nbytes)
    q = rng.integers(0, 256, size=(nq, nbytes), dtype=np.uint8)
    t = rng.integers(0, 256, size=(nt, nbytes), dtype=np.uint8)
    k = min(nq, nt) // 2
    t[:k] = q[:k]                                     # exact matches
    if nt > k + 4:
        t[k:k + 4] = t[0]                             # duplicates: distance ties between train rows
    t[nt // 2, 0] ^= 1
    m = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING)
    idx, dist = m.knnMatch(_dev(torch_mod, q), _dev(torch_mod, t), 2)
    torch_mod.cuda.synchronize()
    widx, wdist = MO.knn2(q, t)
    assert np.array_equal(dist.cpu().numpy(), wdist)
    assert np.array_equal(idx.cpu().numpy(), widx)
    mc = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING, True)
    mm, md = mc.match(_dev(torch_mod, q), _dev(torch_mod, t))
    torch_mod.cuda.synchronize()
    wm, wd = MO.crosscheck(q, t)
    assert np.array_equal(mm.cpu().numpy(), wm) and np.array_equal(md.cpu().numpy(), wd)


def test_matcher_paths_agree(cef, torch_mod, monkeypatch):
    """The popcount kernel and the matrix-core kernel give the same answer on a set with many distance ties.  The knob is
    read when a matcher is created, so both kernels run in this process on the same inputs."""
    rng = np.random.default_rng(99)
    q = rng.integers(0, 256, size=(3000, 64), dtype=np.uint8)
    t = np.concatenate([q[::3], q[::5], rng.integers(0, 256, size=(700, 64), dtype=np.uint8)])     # duplicates: ties everywhere
    monkeypatch.delenv("EFX_MATCH_NO_MFMA", raising=False)
    monkeypatch.delenv("EFX_MATCH_NO_FP4", raising=False)
    m_mfma = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING)          # the FP4 (MX) matrix-core kernel: round 4
    monkeypatch.setenv("EFX_MATCH_NO_FP4", "1")
    m_i8 = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING)            # the int8 matrix-core kernel
    monkeypatch.delenv("EFX_MATCH_NO_FP4")
    monkeypatch.setenv("EFX_MATCH_NO_MFMA", "1")
    m_pop = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING)
    monkeypatch.delenv("EFX_MATCH_NO_MFMA")
    dq, dt = _dev(torch_mod, q), _dev(torch_mod, t)
    idx, dist = m_mfma.knnMatch(dq, dt, 2)
    idx2, dist2 = m_pop.knnMatch(dq, dt, 2)
    idx3, dist3 = m_i8.knnMatch(dq, dt, 2)
    torch_mod.cuda.synchronize()
    assert torch_mod.equal(idx, idx2) and torch_mod.equal(dist, dist2)
    assert torch_mod.equal(idx, idx3) and torch_mod.equal(dist, dist3)
    # 256-bit descriptors through all three kernels as well
    q32, t32 = np.ascontiguousarray(q[:, :32]), np.ascontiguousarray(t[:, :32])
    d32q, d32t = _dev(torch_mod, q32), _dev(torch_mod, t32)
    ra, rb, rc = m_mfma.knnMatch(d32q, d32t, 2), m_pop.knnMatch(d32q, d32t, 2), m_i8.knnMatch(d32q, d32t, 2)
    torch_mod.cuda.synchronize()
    assert torch_mod.equal(ra[0], rb[0]) and torch_mod.equal(ra[1], rb[1]) and torch_mod.equal(ra[0], rc[0]) and torch_mod.equal(ra[1], rc[1])
    from oracle import matcher_oracle as MO
    widx, wdist = MO.knn2(q, t)
    assert np.array_equal(dist.cpu().numpy(), wdist) and np.array_equal(idx.cpu().numpy(), widx)


def test_matcher_on_detected_descriptors(cef, torch_mod):
    """Two views of one scene (shifted crop): cross-checked BAD256 matches are mostly the true correspondences."""
    from oracle import matcher_oracle as MO
    big = synth.synth_frame(700, 900, seed=33)
    a, b = np.ascontiguousarray(big[:600, :800]), np.ascontiguousarray(big[40:640, 60:860])
    det = cef.EfficientFeatures.create(1500, nlevels=1, dtype=cef.EfficientFeatures.BAD_256)
    ka, da = det.detectAndCompute(a)
    kb, db = det.detectAndCompute(b)
    m = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING, True)
    mm, md = m.match(_dev(torch_mod, da), _dev(torch_mod, db))
    torch_mod.cuda.synchronize()
    mm = mm.cpu().numpy()
    wm, _ = MO.crosscheck(da, db)
    assert np.array_equal(mm, wm)
    ok = mm >= 0
    dx = ka["x"][ok] - kb["x"][mm[ok]]
    dy = ka["y"][ok] - kb["y"][mm[ok]]
    good = (np.abs(dx - 60) <= 1) & (np.abs(dy - 40) <= 1)
    assert ok.sum() > 100 and good.mean() > 0.8


def test_cpp_facade_program(cef):
    """The C++ facade (host/efficient_features.hpp) exercised by samples/facade_check.cpp: detectAndCompute, the
    useProvidedKeypoints round trip, mask, uploader + colour conversion, cross-check and knn matcher."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-efficient-features_amd", "efx_facade_check")
    assert os.path.exists(exe), "build it with make -C cuda-efficient-features_amd/csrc"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "facade ok" in r.stdout, r.stdout + r.stderr


def test_context_reuse_across_sizes_and_parameters(cef, torch_mod, oracle):
    """One context, several frames: image size and every parameter change between calls (grow-only scratch buffers and
    the cached pyramid geometry must follow; setDescriptorType swaps the describer, cuda_efficient_features.cpp:373-377)."""
    det = cef.EfficientFeatures.create(2000, dtype=cef.EfficientFeatures.BAD_256)
    steps = [((480, 640), {}),
             ((1080, 1920), dict(nfeatures=6000)),
             ((300, 333), dict(nlevels=4, nonmax_radius=7)),
             ((480, 640), dict(fast_threshold=35, first_level=1, scale_factor=1.35)),
             ((600, 800), dict(desc_type=1))]
    kw = dict(nfeatures=2000, scale_factor=1.2, nlevels=8, first_level=0, fast_threshold=20, nonmax_radius=15, desc_type=0)
    for i, (shape, change) in enumerate(steps):
        kw.update(change)
        det.setMaxFeatures(kw["nfeatures"]); det.setScaleFactor(kw["scale_factor"]); det.setNLevels(kw["nlevels"])
        det.setFirstLevel(kw["first_level"]); det.setFastThreshold(kw["fast_threshold"]); det.setNonmaxRadius(kw["nonmax_radius"])
        det.setDescriptorType(kw["desc_type"])
        assert det.getNLevels() == kw["nlevels"] and det.getDescriptorType() == kw["desc_type"]
        assert det.descriptorSize() == (32 if kw["desc_type"] == 0 else 64)
        img = synth.synth_frame(shape[0], shape[1], seed=70 + i)
        kps, desc, cnt = det.detectAndComputeAsync(_dev(torch_mod, img))
        torch_mod.cuda.synchronize()
        n = int(cnt.item())
        ref = oracle.detect_and_compute(img, **kw)
        assert n == ref["n"], f"step {i}"
        assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32)), f"step {i}"
        assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"]), f"step {i}"


def test_two_threads_two_contexts(cef, torch_mod, oracle):
    """A context is not re-entrant, but contexts are independent (no process-global tables, unlike cuda_bad.cu:49-50):
    two host threads, each with its own context and stream, BAD256 in one and BAD512 in the other."""
    import threading
    imgs = [synth.synth_frame(480, 640, seed=90 + i) for i in range(2)]
    dts = [cef.EfficientFeatures.BAD_256, cef.EfficientFeatures.BAD_512]
    out = [None, None]

    def work(i):
        torch_mod.cuda.set_device(0)
        det = cef.EfficientFeatures.create(3000, dtype=dts[i])
        st = torch_mod.cuda.Stream()
        d_img = _dev(torch_mod, imgs[i])
        res = []
        for _ in range(6):
            kps, desc, cnt = det.detectAndComputeAsync(d_img, stream=st)
            st.synchronize()
            n = int(cnt.item())
            res.append((n, kps[:, :n].cpu().numpy().copy(), desc[:n].cpu().numpy().copy()))
        out[i] = res

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(2):
        ref = oracle.detect_and_compute(imgs[i], nfeatures=3000, desc_type=dts[i])
        for n, k, d in out[i]:
            assert n == ref["n"] and np.array_equal(k.view(np.uint32), ref["kps"].view(np.uint32)) and np.array_equal(d, ref["desc"])


def test_batched_entry_point(cef, torch_mod, oracle):
    """efx_detect_and_compute_batch_async == the per-frame calls (SURVEY 8b "batched variants"): 5 frames over 2 contexts."""
    imgs = [synth.synth_frame(480, 640, seed=120 + i) for i in range(5)]
    d_imgs = [_dev(torch_mod, im) for im in imgs]
    dets = [cef.EfficientFeatures.create(2500, dtype=cef.EfficientFeatures.BAD_256) for _ in range(2)]
    streams = [torch_mod.cuda.Stream() for _ in range(2)]
    kps = [torch_mod.zeros((5, 2500), dtype=torch_mod.float32, device="cuda") for _ in imgs]
    desc = [torch_mod.zeros((2500, 32), dtype=torch_mod.uint8, device="cuda") for _ in imgs]
    cnt = [torch_mod.zeros(1, dtype=torch_mod.int32, device="cuda") for _ in imgs]
    b = cef.Batch(dets, streams, d_imgs, kps, desc, cnt, 2500)
    b.run(); b.run()
    torch_mod.cuda.synchronize()
    for i, im in enumerate(imgs):
        ref = oracle.detect_and_compute(im, nfeatures=2500, desc_type=oracle.BAD_256)
        n = int(cnt[i].item())
        assert n == ref["n"]
        assert np.array_equal(kps[i][:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))
        assert np.array_equal(desc[i][:n].cpu().numpy(), ref["desc"])


@pytest.mark.parametrize("cols,pitch", [(641, 644), (642, 642), (639, 640), (640, 1024)])
def test_widths_and_pitches(cef, torch_mod, oracle, cols, pitch):
    """Frame widths that are not multiples of 4, with aligned and unaligned row pitches: the dword / buffer-resource
    loaders of resize_kernel, fast_kernel and harris_kernel and their byte fallbacks must all give the same answer."""
    img = synth.synth_frame(480, cols, seed=33)
    big = np.zeros((480, pitch), np.uint8)
    big[:, :cols] = img
    d = _dev(torch_mod, big)[:, :cols]
    det = cef.EfficientFeatures.create(3000, dtype=cef.EfficientFeatures.BAD_256)
    kps, desc, cnt = det.detectAndComputeAsync(d)
    torch_mod.cuda.synchronize()
    n = int(cnt.item())
    ref = oracle.detect_and_compute(img, nfeatures=3000, desc_type=oracle.BAD_256)
    assert n == ref["n"]
    assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))
    assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"])


def test_small_quota_and_zero_capacity(cef, torch_mod, oracle):
    img = synth.synth_frame(480, 640, seed=34)
    got, ref = _detect_both(cef, torch_mod, oracle, img, desc_type=0, nfeatures=10)
    _assert_same_keypoints(got, ref)
    assert np.array_equal(got["desc"], ref["desc"])
    det = cef.EfficientFeatures.create(500)
    kps, cnt = det.detectAsync(_dev(torch_mod, img), capacity=0)
    torch_mod.cuda.synchronize()
    assert int(cnt.item()) == 0


def test_mask_with_first_level_and_pitch(cef, torch_mod, oracle):
    img = synth.synth_frame(480, 640, seed=35)
    m = np.zeros((480, 640), np.uint8); m[100:400, 50:600] = 1
    mbig = np.zeros((480, 700), np.uint8); mbig[:, :640] = m
    det = cef.EfficientFeatures.create(3000, 1.2, 8, 2, 20, 15, cef.EfficientFeatures.BAD_256)
    kps, desc, cnt = det.detectAndComputeAsync(_dev(torch_mod, img), mask=_dev(torch_mod, mbig)[:, :640])
    torch_mod.cuda.synchronize()
    n = int(cnt.item())
    ref = oracle.detect_and_compute(img, nfeatures=3000, first_level=2, desc_type=oracle.BAD_256, mask=m)
    assert n == ref["n"] and n > 0
    assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))
    assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"])


@pytest.mark.parametrize("shape,kw", [
    ((480, 640), dict(nfeatures=100000)),                       # quotas above the survivor counts: nothing is cut
    ((480, 640), dict(nfeatures=3000, nlevels=12)),             # upper levels shrink below the 31-px border
    ((240, 320), dict(nfeatures=3000, nlevels=16)),
    ((480, 640), dict(nfeatures=3000, scale_factor=2.0, nlevels=4)),
    ((480, 640), dict(nfeatures=3000, scale_factor=1.05, nlevels=8)),
    ((480, 640), dict(nfeatures=3000, fast_threshold=1)),
    ((480, 640), dict(nfeatures=3000, fast_threshold=200)),
    ((480, 640), dict(nfeatures=3000, nonmax_radius=70)),       # neighbour cells beyond the 3x3 tiles held in LDS
    ((480, 640), dict(nfeatures=3000, nonmax_radius=1)),
    ((31, 31), dict(nfeatures=100)),                            # no pixel inside the border
    ((33, 200), dict(nfeatures=100)),
    ((1, 1), dict(nfeatures=100)),
])
def test_parameter_extremes(cef, torch_mod, oracle, shape, kw):
    img = synth.synth_frame(shape[0], shape[1], seed=55, density=1.0) if min(shape) >= 8 else np.full(shape, 9, np.uint8)
    got, ref = _detect_both(cef, torch_mod, oracle, img, desc_type=0, **kw)
    _assert_same_keypoints(got, ref)
    assert np.array_equal(got["desc"], ref["desc"])


def test_descriptor_keypoint_extremes(cef, oracle):
    """Keypoints outside the image, on its corners, with zero / negative / huge sizes and every angle convention
    (-1 = axis aligned, < 0 = no rotation; bad.cpp:127,138): defined behaviour on both sides, no crash, same bytes."""
    img = synth.synth_frame(200, 260, seed=77, density=2.0)
    kps = np.array([[-50, -50, 31, 10], [400, 10, 31, 0], [130, 100, 0, 0], [130, 100, 90, 45], [0, 0, 31, -1],
                    [259, 199, 31, 359.9], [130, 100, 31, -7], [5.5, 190.25, 17.3, 123.4], [130, 100, 1, 90],
                    [129.99, 99.99, 64, 180]], np.float32)
    for nbits, enum in ((256, cef.BAD.SIZE_256_BITS), (512, cef.BAD.SIZE_512_BITS)):
        assert np.array_equal(cef.BAD.create(1.0, enum).compute(img, kps), oracle.bad_compute(img, kps, nbits))
    got = cef.HashSIFT.create(1.0, cef.HashSIFT.SIZE_256_BITS).compute(img, kps)
    want = oracle.hashsift_compute(img, kps, 256)
    assert np.array_equal(got, want)          # 10 keypoints: nothing may differ (the rate is 2e-5 of the bytes)
    # a keypoint whose window cannot fit the 160 KB LDS is refused, not silently mis-described
    with pytest.raises(cef.EfxError):
        cef.BAD.create(1.0, cef.BAD.SIZE_256_BITS).compute(img, np.array([[100, 100, 5000, 0]], np.float32))


@pytest.mark.parametrize("period,radius", [(9, 15), (12, 8), (17, 16), (6, 3)])
def test_equal_responses_suppress_each_other(cef, torch_mod, oracle, period, radius):
    """A checkerboard gives every crossing the same Harris response: corners of equal strength inside the radius kill
    each other (IsMaxPoint compares with <=, cuda_efficient_features.cu:90), including the strongest corner of a cell
    and its equal twin in the same cell."""
    y, x = np.mgrid[0:300, 0:400]
    img = ((((x // period) + (y // period)) & 1) * 200).astype(np.uint8)
    got, ref = _detect_both(cef, torch_mod, oracle, img, desc_type=1, nonmax_radius=radius, nlevels=4)
    _assert_same_keypoints(got, ref)
    assert np.array_equal(got["desc"], ref["desc"])


@pytest.mark.parametrize("mode", ["tower", "chain_rows", "chain_rows:1,1,1,1,1,1,1", "chain_rows:2,2,2,1", "chain_rows:3,4", "chain_rows:4,3", "chain_rows:1,3,2,1",
                                  "chain_streamed", "chain_plain"])
@pytest.mark.parametrize("shape,scale", [((480, 640), 1.2), ((501, 703), 1.2), ((333, 1111), 1.1), ((700, 900), 1.5), ((600, 800), 2.0),
                                         ((1201, 2504), 1.2), ((997, 1500), 1.7), ((64, 3000), 1.2), ((2100, 300), 1.25)])
def test_pyramid_kernel_variants_bit_exact(cef, torch_mod, oracle, monkeypatch, mode, shape, scale):
    """The four ways a pyramid is produced -- one tower launch (small frames), one to four levels per launch by waves walking down
    strips (large frames, round 5: resize_rows_kernel, with every split of the levels over launches tried here), the streamed
    per-level kernel and the one-tile-per-workgroup per-level kernel (other scale factors, unaligned sources) -- give the same
    levels, bit for bit."""
    if mode != "tower":
        monkeypatch.setenv("EFX_NO_TOWER", "1")
    monkeypatch.delenv("EFX_ROWS_SPLIT", raising=False)
    if mode.startswith("chain_rows:"):
        monkeypatch.setenv("EFX_ROWS_SPLIT", mode.split(":")[1])        # levels per launch of resize_rows_kernel (default: efx_api.cpp)
    if mode in ("chain_streamed", "chain_plain"):
        monkeypatch.setenv("EFX_NO_RESIZE_ROWS", "1")
    if mode == "chain_plain":
        monkeypatch.setenv("EFX_NO_RESIZE_STREAM", "1")
    img = synth.synth_frame(shape[0], shape[1], seed=21)
    det = cef.EfficientFeatures.create(1000, scale, 8, 0, 20, 15, 0)
    d_img = _dev(torch_mod, img)
    det.detectAsync(d_img)
    torch_mod.cuda.synchronize()
    for level in range(8):
        got = det.copyLevel(level, shape[0], shape[1]).cpu().numpy()
        want = oracle.pyramid_level(img, level, scale_factor=scale)
        assert got.shape == want.shape
        assert np.array_equal(got, want), f"{mode} level {level}: {np.count_nonzero(got != want)} pixels differ"
    ref = oracle.detect_and_compute(img, desc_type=-1, nfeatures=1000, scale_factor=scale)
    assert det.lastCount() == ref["n"]


@pytest.mark.parametrize("nlevels,scale", [(17, 1.05), (20, 1.05), (26, 1.04), (32, 1.03)])
def test_more_than_16_levels(cef, torch_mod, oracle, nlevels, scale):
    """nlevels up to EFX_MAX_LEVELS = 32 is accepted (validate_params), so the packed per-tile word must carry 5 level
    bits: with 4, tiles of level >= 16 were decoded as level - 16 (efx_api.cpp tile info word)."""
    img = synth.synth_frame(480, 640, seed=77)
    got, ref = _detect_both(cef, torch_mod, oracle, img, desc_type=1, nfeatures=3000, nlevels=nlevels, scale_factor=scale)
    _assert_same_keypoints(got, ref)
    assert np.array_equal(got["desc"], ref["desc"])
    assert sum(ref["stats"]["n_kept"][16:]) > 0          # the upper levels really contribute keypoints


@pytest.mark.parametrize("desc_type", [0, 1, 3])
def test_quota_sum_exceeds_nfeatures(cef, torch_mod, oracle, desc_type):
    """calcNumFeaturesPerLevel (.cpp:159-174) rounds every level's quota up, so the detector can emit more than nfeatures
    keypoints (nfeatures 3, 5 levels, scale 1.01: quotas 1,1,1,1,0).  Every emitted keypoint must be described, as in the
    reference (the describe launch was bounded by nfeatures)."""
    img = synth.synth_frame(300, 400, seed=5)
    det = cef.EfficientFeatures.create(3, 1.01, 5, 0, 20, 15, desc_type)
    d_img = _dev(torch_mod, img)
    cap = 16
    nbytes = det.descriptorSize()
    kps = torch_mod.zeros((5, cap), dtype=torch_mod.float32, device="cuda")
    desc = torch_mod.full((cap, nbytes), 0xEE, dtype=torch_mod.uint8, device="cuda")
    _, _, cnt = det.detectAndComputeAsync(d_img, kps, desc, capacity=cap)
    torch_mod.cuda.synchronize()
    n = int(cnt.item())
    ref = oracle.detect_and_compute(img, nfeatures=3, scale_factor=1.01, nlevels=5, desc_type=desc_type, capacity=cap)
    assert n == ref["n"] == 4
    assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))
    if desc_type < 2:
        assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"])
    else:
        # four HashSIFT descriptors: at most ONE rounding event (a few bytes of one descriptor; measured rate 2e-5 of the bytes)
        g = desc[:n].cpu().numpy()
        rows_off = np.nonzero((g != ref["desc"]).any(axis=1))[0]
        assert rows_off.size <= 1 and np.count_nonzero(g != ref["desc"]) <= 4 and not np.all(desc[3].cpu().numpy() == 0xEE)


def test_detect_and_compute_without_descriptors(cef, torch_mod, oracle):
    """detectAndComputeAsync(want_descriptors=False) is the detect-only mode of that entry point."""
    img = synth.synth_frame(240, 320, seed=9)
    det = cef.EfficientFeatures.create(500, dtype=cef.EfficientFeatures.BAD_256)
    kps, desc, cnt = det.detectAndComputeAsync(_dev(torch_mod, img), want_descriptors=False)
    torch_mod.cuda.synchronize()
    ref = oracle.detect_and_compute(img, nfeatures=500, desc_type=-1)
    n = int(cnt.item())
    assert desc is None and n == ref["n"]
    assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))


# ---- whole-level blur (round 4): blur_levels_kernel + a wave per keypoint, against the per-window blur and the oracle ----
@pytest.mark.parametrize("cols", [1040, 1041, 1042, 1043, 700, 509, 512, 513, 768, 769])
def test_level_blur_paths_equal_oracle(cef, torch_mod, oracle, cols, monkeypatch):
    """detectAndCompute BAD behind the whole-level Gaussian: levels of 512 columns and more take the dword path of
    blur_levels_kernel, whose right-edge strip patches the BORDER_REFLECT_101 pixels by r = cols - ((cols - 1) & ~3) (1 .. 4:
    every residue of cols modulo 4, here and again on the upper levels; 509 .. 513 and 768 / 769: the anchored strip starting exactly on
    / next to a regular strip's boundary), narrower levels the byte path; an image whose base is
    not 4-byte aligned sends level 0 through the byte path as well.  All of them, and the per-window blur (EFX_NO_LEVEL_BLUR),
    must give the oracle's keypoints and descriptor bytes."""
    torch = torch_mod
    rows = 560
    img = synth.synth_frame(rows, cols, seed=900 + cols, density=1.5)
    ref = oracle.detect_and_compute(img, nfeatures=6000, nonmax_radius=7, desc_type=oracle.BAD_512)
    big = torch.zeros((rows, cols + 61), dtype=torch.uint8, device="cuda")
    for off in (0, 1):                                      # aligned base, then base + 1 with an odd pitch
        view = big[:, off:off + cols]
        view.copy_(torch.from_numpy(img).cuda())
        for knob in (False, True):
            if knob:
                monkeypatch.setenv("EFX_NO_LEVEL_BLUR", "1")
            else:
                monkeypatch.delenv("EFX_NO_LEVEL_BLUR", raising=False)
            det = cef.EfficientFeatures.create(6000, 1.2, 8, 0, 20, 7, cef.EfficientFeatures.BAD_512)     # knobs are read here
            kps, desc, cnt = det.detectAndComputeAsync(view)
            torch.cuda.synchronize()
            n = int(cnt.item())
            assert n == ref["n"] and n > 1500, (off, knob, n)
            assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32)), (off, knob)
            bad = np.nonzero((desc[:n].cpu().numpy() != ref["desc"]).any(axis=1))[0]
            assert bad.size == 0, f"offset {off}, EFX_NO_LEVEL_BLUR {knob}: {bad.size} descriptors differ, first at keypoints {bad[:6]}"
    monkeypatch.delenv("EFX_NO_LEVEL_BLUR", raising=False)


@pytest.mark.parametrize("kind", ["flat", "half_flat", "smooth"])
def test_frames_with_empty_tiles(cef, torch_mod, oracle, kind):
    """Tiles without FAST corners leave harris_kernel / nms_kernel at once (round 5): a constant frame (no corner anywhere), a frame
    whose lower half is constant (empty tiles beside full ones, at every level) and a 1/f^2 frame (nothing above the threshold)."""
    torch = torch_mod
    if kind == "flat":
        img = np.full((1080, 1920), 128, np.uint8)
    elif kind == "half_flat":
        img = np.concatenate([synth.synth_frame(540, 1920, seed=5), np.full((540, 1920), 60, np.uint8)])
    else:
        img = synth.powerlaw_frame(720, 1280, seed=3, beta=2.0, contrast=40.0)
    img = np.ascontiguousarray(img)
    det = cef.EfficientFeatures.create(5000, dtype=cef.EfficientFeatures.BAD_256)
    kps, desc, cnt = det.detectAndComputeAsync(torch.from_numpy(img).cuda()); torch.cuda.synchronize()
    n = int(cnt.item())
    ref = oracle.detect_and_compute(img, nfeatures=5000, desc_type=oracle.BAD_256)
    assert n == ref["n"]
    assert (n > 0) == (kind == "half_flat")
    assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))
    assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"])


def test_detect_and_compute_is_graph_capturable(cef, torch_mod, oracle):
    """The whole detectAndCompute launch sequence (17 kernels, no host synchronisation, no allocation once the context has seen
    the geometry) can be captured into a HIP graph and replayed: same keypoints and descriptors as the direct call and as the
    oracle, also for a DIFFERENT image written into the captured input buffer.  (Replay does not shorten a call -- the
    ~4.5 us between dependent kernels is the GPU's, tools/microbench/graph_latency.py -- but it takes the 55 us of host
    enqueue work off the caller's thread.)"""
    torch = torch_mod
    a = synth.synth_frame(480, 640, seed=61)
    b = synth.synth_frame(480, 640, seed=62, density=1.0)
    img = torch.from_numpy(a).cuda()
    det = cef.EfficientFeatures.create(3000, dtype=cef.EfficientFeatures.BAD_256)
    kps = torch.zeros((5, 3000), dtype=torch.float32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    desc = torch.zeros((3000, 32), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        det.detectAndComputeAsync(img, kps, desc, cnt, stream=s)          # geometry, arenas, side buffers: allocated here
        s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        det.detectAndComputeAsync(img, kps, desc, cnt, stream=s)
    for frame in (a, b, a):
        img.copy_(torch.from_numpy(frame).cuda())
        kps.zero_(); desc.zero_(); cnt.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ref = oracle.detect_and_compute(frame, nfeatures=3000, desc_type=oracle.BAD_256)
        n = int(cnt.item())
        assert n == ref["n"]
        assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))
        assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"])


@pytest.mark.parametrize("fork", ["auto", "1", "2", "3"])
def test_level_blur_side_stream_is_chosen_per_call(cef, torch_mod, oracle, monkeypatch, fork):
    """Where the level blur of detectAndCompute (BAD) runs is decided per call (efx_api.cpp, detect_common): on the context's
    side stream when the call's stream had nothing pending at this call and at the one before -- a caller that waits for
    every frame -- and on the call's stream otherwise.  Same keypoints and descriptors either way, also when the two kinds of call
    alternate on one context, on two user streams, and for different frames in the same buffers.
    The per-call decision only applies to pyramids of >= 50 M pixels; EFX_BLUR_FORK_MIN_PX (read with the other knobs when a context
    is created -- ADVICE r5: it used to be a process-wide static set at module import, which changed the path of every other GPU test
    module collected with this one) lowers the gate so that these 600 x 800 frames take it ("auto"), and EFX_BLUR_FORK = 1 / 2 /
    3 (read when a context is created) forces the fork behind the pyramid / harris_kernel / nms_kernel on every call: fork / join
    events reused across streams, the side stream shared by consecutive calls."""
    torch = torch_mod
    monkeypatch.setenv("EFX_BLUR_FORK_MIN_PX", "400000")
    if fork == "auto":
        monkeypatch.delenv("EFX_BLUR_FORK", raising=False)
    else:
        monkeypatch.setenv("EFX_BLUR_FORK", fork)
    EF = cef.EfficientFeatures
    frames = [synth.synth_frame(600, 800, seed=71 + i, density=0.5 + 0.25 * i) for i in range(3)]
    refs = [oracle.detect_and_compute(f, nfeatures=4000, desc_type=oracle.BAD_512) for f in frames]
    det = EF.create(4000, dtype=EF.BAD_512)
    img = torch.zeros((600, 800), dtype=torch.uint8, device="cuda")
    kps = torch.zeros((5, 4000), dtype=torch.float32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    desc = torch.zeros((4000, 64), dtype=torch.uint8, device="cuda")

    def check(i):
        n = int(cnt.item())
        assert n == refs[i]["n"]
        assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), refs[i]["kps"].view(np.uint32))
        assert np.array_equal(desc[:n].cpu().numpy(), refs[i]["desc"])

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    # call, wait, call, wait ...: from the second call on the blur runs on the side stream
    for rep in range(6):
        i = rep % 3
        with torch.cuda.stream(s1):
            img.copy_(torch.from_numpy(frames[i]).cuda(), non_blocking=False)
            s1.synchronize()
            det.detectAndComputeAsync(img, kps, desc, cnt, stream=s1)
        s1.synchronize()
        check(i)
    # back to back without waiting (the stream is busy: inline), then waiting again, on another stream
    outs = [(torch.zeros_like(kps), torch.zeros_like(desc), torch.zeros_like(cnt)) for _ in range(6)]
    imgs = [torch.from_numpy(frames[j % 3]).cuda() for j in range(6)]
    torch.cuda.synchronize()
    for j in range(6):
        det.detectAndComputeAsync(imgs[j], outs[j][0], outs[j][1], outs[j][2], stream=s2)
    s2.synchronize()
    for j in range(6):
        kps, desc, cnt = outs[j]
        check(j % 3)
    for j in range(4):
        det.detectAndComputeAsync(imgs[j], outs[j][0], outs[j][1], outs[j][2], stream=s2 if j % 2 else s1)
        torch.cuda.synchronize()
        kps, desc, cnt = outs[j]
        check(j % 3)


def test_matcher_kernels_agree_on_tie_heavy_random_sets(cef, torch_mod, monkeypatch):
    """The three Hamming kernels (FP4 matrix cores, int8 matrix cores, popcount) on random set sizes and descriptors that
    produce ties everywhere: exact duplicates from a small pool, descriptors with few set bits (a handful of distinct
    distances), noisy copies.  Such sets make EVERY best-two update matter: round 4's first group filter (an inline-asm
    v_max3_f32 the compiler could not see reading the MFMA's result registers) lost about one update in 300 on them while every
    uniformly random set of this module still passed.  tools/microbench/match_fuzz.py is the long form."""
    torch = torch_mod
    from oracle import matcher_oracle as MO
    for k in ("EFX_MATCH_NO_MFMA", "EFX_MATCH_NO_FP4"):
        monkeypatch.delenv(k, raising=False)
    m_fp4 = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING)
    monkeypatch.setenv("EFX_MATCH_NO_FP4", "1")
    m_i8 = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING)
    monkeypatch.delenv("EFX_MATCH_NO_FP4")
    monkeypatch.setenv("EFX_MATCH_NO_MFMA", "1")
    m_pop = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING)
    monkeypatch.delenv("EFX_MATCH_NO_MFMA")
    rng = np.random.default_rng(2024)
    for case in range(48):
        nb = int(rng.choice([32, 64]))
        nq, nt = int(rng.integers(128, 2500)), int(rng.integers(64, 4000))
        kind = case % 4
        if kind == 0:
            q = rng.integers(0, 256, size=(nq, nb), dtype=np.uint8); t = rng.integers(0, 256, size=(nt, nb), dtype=np.uint8)
        elif kind == 1:
            pool = rng.integers(0, 256, size=(int(rng.integers(2, 40)), nb), dtype=np.uint8)
            q = pool[rng.integers(0, len(pool), nq)]; t = pool[rng.integers(0, len(pool), nt)]
        elif kind == 2:
            q = (rng.random((nq, nb)) < 0.03).astype(np.uint8) * np.uint8(1 << int(rng.integers(0, 8)))
            t = (rng.random((nt, nb)) < 0.03).astype(np.uint8) * np.uint8(1 << int(rng.integers(0, 8)))
        else:
            q = rng.integers(0, 256, size=(nq, nb), dtype=np.uint8)
            t = q[rng.integers(0, nq, nt)] ^ ((rng.random((nt, nb)) < 0.02).astype(np.uint8) * np.uint8(16))
        q, t = np.ascontiguousarray(q), np.ascontiguousarray(t)
        dq, dt = _dev(torch, q), _dev(torch, t)
        a, b, d = m_fp4.knnMatch(dq, dt, 2), m_i8.knnMatch(dq, dt, 2), m_pop.knnMatch(dq, dt, 2)
        torch.cuda.synchronize()
        info = (case, kind, nb, nq, nt)
        assert torch.equal(a[0], d[0]) and torch.equal(a[1], d[1]), ("fp4 vs popcount", info)
        assert torch.equal(b[0], d[0]) and torch.equal(b[1], d[1]), ("int8 vs popcount", info)
        if nq * nt <= 3_000_000:
            widx, wdist = MO.knn2(q, t)
            assert np.array_equal(d[0].cpu().numpy(), widx) and np.array_equal(d[1].cpu().numpy(), wdist), ("popcount vs oracle", info)


@pytest.mark.parametrize("kind", ["sparse", "dense", "noise_bitmap_tiles", "capped", "odd_size"])
def test_packed_harris_kernel_equals_oracle(cef, torch_mod, oracle, monkeypatch, kind):
    """harris_packed_kernel (round 6: several tiles per wave for frames with the corner statistics of photographs; the host takes
    it by the previous frame's density on large frames, EFX_PACK=1 forces it): same records at the same places as harris_kernel --
    sparse groups side by side in one round, dense ones over several, groups with a bitmap tile (more than 256 corners) tile by
    tile, the 10 % cap cutting inside a group, a level whose tile count is not a multiple of the group."""
    monkeypatch.setenv("EFX_PACK", "1")
    kw = {}
    if kind == "sparse":
        img = synth.powerlaw_frame(700, 1000, seed=3, beta=1.3, contrast=45.0)
    elif kind == "dense":
        img = synth.synth_frame(600, 900, seed=77, density=3.0)
    elif kind == "noise_bitmap_tiles":
        img = synth.noise_frame(500, 700, seed=5); kw = dict(fast_threshold=5)
    elif kind == "capped":
        img = synth.noise_frame(640, 640, seed=6)
    else:
        img = synth.synth_frame(333, 517, seed=9)
    got, ref = _detect_both(cef, torch_mod, oracle, img, desc_type=0, nfeatures=3000, **kw)
    _assert_same_keypoints(got, ref)
    assert np.array_equal(got["desc"], ref["desc"])


@pytest.mark.parametrize("t", [0, 1, 2, 3, 4, 5, 6, 7, 19, 21, 22, 23, 63, 64, 127, 128, 129, 200, 251, 252, 253, 254, 255])
def test_fast_quick_test_keeps_every_corner_at_any_threshold(cef, torch_mod, oracle, t):
    """fast_kernel's quick test compares the pixels' upper six bits (round 6: four pixels per 32-bit add; tq = (t + 1) >> 2 is the
    largest coarse threshold that still keeps every pixel the exact compass test keeps).  Every residue of t modulo 4, the
    thresholds around the field boundaries (63 / 64, 127 / 128) and the largest ones, on an image whose pixels use the whole byte
    range -- uniform noise with saturated blobs -- so that corners exist up to t = 254: the same corners as the oracle's exact
    test, level by level."""
    rng = np.random.default_rng(41)
    img = rng.integers(0, 256, size=(300, 420), dtype=np.uint8)
    yy, xx = np.mgrid[0:300, 0:420]
    for k in range(60):                                   # saturated discs and squares on the noise: 0 / 255 corners
        cy, cx, r = int(rng.integers(20, 280)), int(rng.integers(20, 400)), int(rng.integers(3, 9))
        v = 255 if k % 2 else 0
        if k % 3: img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = v
        else: img[cy - r:cy + r, cx - r:cx + r] = v
    got, ref = _detect_both(cef, torch_mod, oracle, img, desc_type=0, nfeatures=20000, fast_threshold=t, nlevels=3)
    _assert_same_keypoints(got, ref)
    assert np.array_equal(got["desc"], ref["desc"])
    if t <= 200:
        assert sum(s["n_candidates"] for s in got["stats"]) > 0
