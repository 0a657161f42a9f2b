"""the arguments the reference accepts but ignores (`mask`, cuda_efficient_features.cpp:225-250) or
asserts away (`useProvidedKeypoints`, :229).  Behaviour is fixed by DESIGN.md specs S12 / S13 and restated in the
oracle; the GPU tests compare the HIP path with it bit for bit and check the size-independent properties."""
import numpy as np
import pytest

from tools import synth


def some_mask(rows, cols, seed):
    rng = np.random.default_rng(seed)
    m = np.zeros((rows, cols), np.uint8)
    for _ in range(6):
        y, x = int(rng.integers(0, rows - 40)), int(rng.integers(0, cols - 40))
        m[y:y + int(rng.integers(30, rows // 2)), x:x + int(rng.integers(30, cols // 2))] = int(rng.integers(1, 256))
    return m


def test_oracle_mask_semantics(oracle):
    img = synth.synth_frame(300, 400, seed=3)
    m = some_mask(300, 400, seed=1)
    r = oracle.detect_and_compute(img, nfeatures=3000, mask=m)
    k = oracle.unpack_keypoints(r["kps"])
    assert r["n"] > 20 and (m[k["y"], k["x"]] != 0).all()                      # every reported keypoint is on the mask
    full = oracle.detect_and_compute(img, nfeatures=3000)
    ones = oracle.detect_and_compute(img, nfeatures=3000, mask=np.full_like(img, 7))
    assert np.array_equal(full["kps"].view(np.uint32), ones["kps"].view(np.uint32))
    assert oracle.detect_and_compute(img, nfeatures=3000, mask=np.zeros_like(img))["n"] == 0
    # masked corners do not suppress their neighbours: a level's survivors inside the mask are a superset of the
    # unmasked run's survivors that lie deep inside the mask
    assert r["stats"]["n_candidates"][0] < full["stats"]["n_candidates"][0]


@pytest.mark.parametrize("dt", [0, 1, 2, 3])
def test_oracle_provided_round_trip(oracle, dt):
    img = synth.synth_frame(300, 400, seed=4)
    r = oracle.detect_and_compute(img, nfeatures=1500, desc_type=dt)
    assert r["n"] > 50
    assert np.array_equal(oracle.compute_provided(img, r["kps"], dt), r["desc"])
    # an out-of-range octave yields a zero descriptor, the others are unaffected
    k = r["kps"].copy()
    k[3, 0] = np.array([99], np.int32).view(np.float32)[0]
    d = oracle.compute_provided(img, k, dt)
    assert not d[0].any() and np.array_equal(d[1:], r["desc"][1:])


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def cef():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cef_loader
    return cef_loader.load()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,seed", [((480, 640), 2), ((300, 1000), 5)])
def test_masked_detect_bit_exact(cef, oracle, shape, seed):
    import torch
    img = synth.synth_frame(shape[0], shape[1], seed=seed)
    m = some_mask(shape[0], shape[1], seed=seed)
    det = cef.EfficientFeatures.create(4000, dtype=cef.EfficientFeatures.BAD_256)
    kps, desc, cnt = det.detectAndComputeAsync(torch.from_numpy(img).cuda(), mask=torch.from_numpy(m).cuda())
    torch.cuda.synchronize()
    n = int(cnt.item())
    ref = oracle.detect_and_compute(img, nfeatures=4000, desc_type=oracle.BAD_256, mask=m)
    assert n == ref["n"] and n > 20
    assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))
    assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"])
    k = oracle.unpack_keypoints(kps[:, :n].cpu().numpy())
    assert (m[k["y"], k["x"]] != 0).all()
    # detect only (no descriptors) with a mask, and the degenerate masks
    k2, _, c2 = det.detectAndComputeAsync(torch.from_numpy(img).cuda(), mask=torch.from_numpy(m).cuda(), want_descriptors=False)
    k0, _, c0 = det.detectAndComputeAsync(torch.from_numpy(img).cuda(), mask=torch.zeros(shape, dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    assert int(c2.item()) == n and np.array_equal(k2[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))
    assert int(c0.item()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [0, 1, 2, 3])
def test_provided_keypoints_round_trip(cef, oracle, dt):
    """detectAndCompute -> (kps, desc); detectAndCompute(kps, useProvidedKeypoints) -> the same desc (spec S13)."""
    import torch
    img = synth.synth_frame(600, 800, seed=9)
    d_img = torch.from_numpy(img).cuda()
    det = cef.EfficientFeatures.create(5000, dtype=dt)
    kps, desc, cnt = det.detectAndComputeAsync(d_img)
    torch.cuda.synchronize()
    n = int(cnt.item())
    assert n > 200
    again = det.detectAndComputeAsync(d_img, keypoints=kps, n=n, useProvidedKeypoints=True)
    torch.cuda.synchronize()
    assert np.array_equal(again.cpu().numpy(), desc[:n].cpu().numpy())
    want = oracle.compute_provided(img, kps[:, :n].cpu().numpy(), dt)
    if dt <= 1:
        assert np.array_equal(again.cpu().numpy(), want)
    else:
        assert np.count_nonzero(again.cpu().numpy() != want) <= max(1, int(1e-4 * want.size))


@pytest.mark.gpu
def test_host_detect_and_compute_ex(cef, oracle):
    img = synth.synth_frame(480, 640, seed=12)
    m = some_mask(480, 640, seed=3)
    det = cef.EfficientFeatures.create(3000, dtype=cef.EfficientFeatures.BAD_512)
    kps, desc = det.detectAndCompute(img, mask=m)
    ref = oracle.detect_and_compute(img, nfeatures=3000, desc_type=oracle.BAD_512, mask=m)
    assert len(kps) == ref["n"] and np.array_equal(desc, ref["desc"])
    kps2, desc2 = det.detectAndCompute(img, useProvidedKeypoints=True, keypoints=kps)
    assert len(kps2) == len(kps) and np.array_equal(desc2, desc)
    bad_oct = kps.copy(); bad_oct["octave"][:3] = 40
    _, d3 = det.detectAndCompute(img, useProvidedKeypoints=True, keypoints=bad_oct)
    assert not d3[:3].any() and np.array_equal(d3[3:], desc[3:])
    with pytest.raises(cef.EfxError):
        det.detectAndCompute(img, mask=np.zeros((10, 10), np.uint8))
