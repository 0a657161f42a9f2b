"""Input stage: BGR/BGRA -> gray and the double-buffered upload.

Reference behaviour: cv::cvtColor(COLOR_BGR2GRAY / COLOR_BGRA2GRAY) in the CPU describers (bad.cpp:268-281,
hash_sift.cpp:51-66) and samples (sample_common.cpp:35-45); getInputMat upload (cuda_efficient_features.cpp:71-84).
cvtColor is third-party arithmetic (spec S11 fixes it: OpenCV's 8-bit fixed-point form)."""
import numpy as np
import pytest

from tools import synth


def numpy_gray(img):
    i = img.astype(np.int64)
    return ((3735 * i[..., 0] + 19235 * i[..., 1] + 9798 * i[..., 2] + 16384) >> 15).astype(np.uint8)


def colour_frame(rows, cols, ch, seed):
    rng = np.random.default_rng(seed)
    base = synth.synth_frame(rows, cols, seed=seed).astype(np.int16)
    img = np.stack([np.clip(base + rng.integers(-40, 40, base.shape), 0, 255) for _ in range(3)], axis=-1).astype(np.uint8)
    if ch == 4:
        img = np.concatenate([img, rng.integers(0, 256, (rows, cols, 1), dtype=np.uint8)], axis=-1)
    return np.ascontiguousarray(img)


@pytest.mark.parametrize("ch", [3, 4])
def test_oracle_gray_equals_numpy_spec(oracle, ch):
    img = colour_frame(37, 53, ch, seed=3)
    assert np.array_equal(oracle.bgr2gray(img), numpy_gray(img))
    # known answers of the fixed-point form: white -> 255, pure channels, mid gray exact
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128], [0, 0, 0]]], np.uint8)
    assert oracle.bgr2gray(px).tolist() == [[255, 29, 150, 76, 128, 0]]


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def cef():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cef_loader
    return cef_loader.load()


@pytest.mark.gpu
@pytest.mark.parametrize("ch", [3, 4])
@pytest.mark.parametrize("shape", [(480, 640), (101, 333), (7, 5)])
def test_cvt_gray_bit_exact(cef, oracle, ch, shape):
    import torch
    img = colour_frame(shape[0], shape[1], ch, seed=11)
    got = cef.cvtGray(torch.from_numpy(img).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), oracle.bgr2gray(img))


@pytest.mark.gpu
def test_describer_accepts_colour_host_images(cef, oracle):
    """cv::BAD::compute / cv::HashSIFT::compute take 8UC3 / 8UC4 and convert first (bad.cpp:268-281)."""
    img = colour_frame(240, 320, 3, seed=5)
    img4 = np.concatenate([img, np.full((240, 320, 1), 9, np.uint8)], axis=-1)
    kps = synth.random_keypoints(240, 320, 300, seed=2)
    want = oracle.bad_compute(oracle.bgr2gray(img), kps, 256)
    bad = cef.BAD.create(1.0, cef.BAD.SIZE_256_BITS)
    assert np.array_equal(bad.compute(img, kps), want)
    assert np.array_equal(bad.compute(img4, kps), want)
    with pytest.raises(cef.EfxError):
        bad.compute(np.zeros((10, 10, 2), np.uint8), kps)


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [False, True])
def test_uploader_ring_matches_direct_path(cef, oracle, pinned):
    """Five frames (gray, BGR, BGRA mixed) through the two-slot uploader: every frame's keypoints and descriptors
    equal the oracle's on the converted frame, i.e. no slot is overwritten while it is still being read."""
    import ctypes
    import torch
    up = cef.Uploader()
    det = cef.EfficientFeatures.create(2000, dtype=cef.EfficientFeatures.BAD_256)
    stream = torch.cuda.Stream()
    frames, outs, keep = [], [], []
    for k in range(5):
        ch = (1, 3, 4, 3, 1)[k]
        src = synth.synth_frame(300, 400, seed=40 + k) if ch == 1 else colour_frame(300, 400, ch, seed=40 + k)
        if pinned:
            h = cef.host_alloc(src.shape); h[...] = src; keep.append(h)
        else:
            h = src
        frames.append(src)
        d_ptr, pitch, rows, cols = up.upload(h, stream=stream)
        kps = torch.zeros((5, 2000), dtype=torch.float32, device="cuda")
        desc = torch.zeros((2000, 32), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        rc = cef.lib().efx_detect_and_compute_async(det._h, ctypes.c_void_p(d_ptr), rows, cols, ctypes.c_size_t(pitch),
                                                    ctypes.c_void_p(kps.data_ptr()), ctypes.c_size_t(kps.stride(0) * 4),
                                                    ctypes.c_void_p(desc.data_ptr()), ctypes.c_size_t(32), 2000,
                                                    ctypes.c_void_p(cnt.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0
        outs.append((kps, desc, cnt))
    stream.synchronize()
    for src, (kps, desc, cnt) in zip(frames, outs):
        gray = src if src.ndim == 2 else oracle.bgr2gray(src)
        ref = oracle.detect_and_compute(gray, nfeatures=2000, desc_type=oracle.BAD_256)
        n = int(cnt.item())
        assert n == ref["n"]
        assert np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref["kps"].view(np.uint32))
        assert np.array_equal(desc[:n].cpu().numpy(), ref["desc"])
    for h in keep:
        cef.host_free(h)
