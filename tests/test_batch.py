"""Frame-batched launches (round 6; SURVEY 8b "batched variants (..., nframes) for roofline-sized launches"): the frames a
context owns in efx_detect_and_compute_batch_async go through ONE launch of every kernel (frame = blockIdx.y).  Every frame
of a batch must equal the single-frame call on the same image bit for bit (keypoint rows and descriptor bytes) -- and, where
the oracle is fast enough, the oracle.  Reference loop this replaces: samples/sample_image_sequence.cpp:70-105."""
import os

import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu


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


def _single(cef, torch, d_img, nfeatures, dtype, describe=True, **kw):
    det = cef.EfficientFeatures.create(nfeatures, kw.get("scale_factor", 1.2), kw.get("nlevels", 8), kw.get("first_level", 0),
                                       kw.get("fast_threshold", 20), kw.get("nonmax_radius", 15), dtype)
    if describe:
        kps, desc, cnt = det.detectAndComputeAsync(d_img)
    else:
        kps, cnt = det.detectAsync(d_img)
        desc = None
    torch.cuda.synchronize()
    n = int(cnt.item())
    return n, kps[:, :n].cpu().numpy().view(np.uint32), None if desc is None else desc[:n].cpu().numpy()


def _run_batch(cef, torch, d_imgs, nfeatures, dtype, nctx=1, describe=True, runs=1, **kw):
    dets = [cef.EfficientFeatures.create(nfeatures, kw.get("scale_factor", 1.2), kw.get("nlevels", 8), kw.get("first_level", 0),
                                         kw.get("fast_threshold", 20), kw.get("nonmax_radius", 15), dtype) for _ in range(nctx)]
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    nbytes = dets[0].descriptorSize()
    # poisoned outputs: a frame the batch does not write shows up
    kps = [torch.full((5, nfeatures), -7.0, dtype=torch.float32, device="cuda") for _ in d_imgs]
    desc = [torch.full((nfeatures, nbytes), 0xA5, dtype=torch.uint8, device="cuda") for _ in d_imgs] if describe else None
    cnt = [torch.full((1,), -1, dtype=torch.int32, device="cuda") for _ in d_imgs]
    torch.cuda.synchronize()
    b = cef.Batch(dets, streams, d_imgs, kps, desc, cnt, nfeatures)
    for _ in range(runs):
        b.run()
    torch.cuda.synchronize()
    out = []
    for i in range(len(d_imgs)):
        n = int(cnt[i].item())
        out.append((n, kps[i][:, :n].cpu().numpy().view(np.uint32), None if desc is None else desc[i][:n].cpu().numpy()))
    return out, dets


def _same(a, b, what):
    assert a[0] == b[0], f"{what}: N {a[0]} != {b[0]}"
    assert np.array_equal(a[1], b[1]), f"{what}: keypoint rows differ"
    if a[2] is not None:
        assert np.array_equal(a[2], b[2]), f"{what}: descriptor bytes differ"


def test_batch_of_16_equals_oracle(cef, torch_mod, oracle):
    """EFX_MAX_BATCH frames on ONE context (one launch of every kernel), BAD512, each frame against the oracle."""
    imgs = [synth.synth_frame(360, 480, seed=500 + i, density=[0.3, 1.0, 3.0][i % 3]) for i in range(16)]
    d = [_dev(torch_mod, im) for im in imgs]
    out, dets = _run_batch(cef, torch_mod, d, 1500, cef.EfficientFeatures.BAD_512, runs=2)
    for i, im in enumerate(imgs):
        ref = oracle.detect_and_compute(im, nfeatures=1500, desc_type=oracle.BAD_512)
        _same(out[i], (ref["n"], ref["kps"].view(np.uint32), ref["desc"]), f"frame {i}")
    # the context's summary is that of the batch's last frame
    assert dets[0].lastCount() == out[-1][0]


@pytest.mark.parametrize("dtype_name", ["BAD_256", "HASH_SIFT_256", "HASH_SIFT_512"])
def test_batch_equals_single_frame_calls(cef, torch_mod, dtype_name):
    """Other describers behind a batched detect (HashSIFT: one describe per frame on the frame's buffers)."""
    dtype = getattr(cef.EfficientFeatures, dtype_name)
    imgs = [synth.synth_frame(540, 720, seed=700 + i) for i in range(5)]
    d = [_dev(torch_mod, im) for im in imgs]
    out, _ = _run_batch(cef, torch_mod, d, 3000, dtype)
    for i in range(len(imgs)):
        _same(out[i], _single(cef, torch_mod, d[i], 3000, dtype)[:3], f"{dtype_name} frame {i}")


def test_batch_detect_only_and_chunks_of_16(cef, torch_mod):
    """No descriptor matrices (detect only), and 19 frames on one context: a launch of 16 and one of 3."""
    imgs = [synth.synth_frame(240, 320, seed=900 + i) for i in range(19)]
    d = [_dev(torch_mod, im) for im in imgs]
    out, _ = _run_batch(cef, torch_mod, d, 800, cef.EfficientFeatures.BAD_256, describe=False)
    for i in range(len(imgs)):
        _same(out[i], _single(cef, torch_mod, d[i], 800, cef.EfficientFeatures.BAD_256, describe=False)[:3], f"frame {i}")


def test_batch_4k_rows_chain_and_three_contexts(cef, torch_mod):
    """4K frames take the row-walking pyramid chain (resize_rows_kernel) and one wave per tile; 5 frames over 3 contexts."""
    imgs = [synth.synth_frame(2160, 3840, seed=1200 + i) for i in range(5)]
    d = [_dev(torch_mod, im) for im in imgs]
    out, _ = _run_batch(cef, torch_mod, d, 20000, cef.EfficientFeatures.BAD_512, nctx=3, runs=2)
    for i in range(len(imgs)):
        _same(out[i], _single(cef, torch_mod, d[i], 20000, cef.EfficientFeatures.BAD_512)[:3], f"4K frame {i}")


def test_batch_with_one_unaligned_frame(cef, torch_mod):
    """Alignment decisions about level 0 (dword staging, LDS-DMA) are taken once per launch: ONE frame whose base address is odd
    sends the whole batch through the byte paths, with the same results."""
    imgs = [synth.synth_frame(480, 644, seed=1300 + i) for i in range(4)]
    d = [_dev(torch_mod, im) for im in imgs]
    flat = torch_mod.zeros(480 * 644 + 8, dtype=torch_mod.uint8, device="cuda")
    odd = flat[1:1 + 480 * 644].view(480, 644)
    odd.copy_(d[2])
    assert odd.data_ptr() % 4 != 0
    d2 = [d[0], d[1], odd, d[3]]
    out, _ = _run_batch(cef, torch_mod, d2, 2000, cef.EfficientFeatures.BAD_256)
    for i in range(4):
        _same(out[i], _single(cef, torch_mod, d[i], 2000, cef.EfficientFeatures.BAD_256)[:3], f"frame {i}")


def test_batch_argument_checks(cef, torch_mod):
    """A batch either has a keypoint matrix for every frame or for none; frames are never silently skipped."""
    import ctypes as C
    det = cef.EfficientFeatures.create(500, dtype=cef.EfficientFeatures.BAD_256)
    imgs = [_dev(torch_mod, synth.synth_frame(240, 320, seed=5 + i)) for i in range(2)]
    kps = [torch_mod.zeros((5, 500), dtype=torch_mod.float32, device="cuda") for _ in imgs]
    cnt = [torch_mod.zeros(1, dtype=torch_mod.int32, device="cuda") for _ in imgs]
    P = C.c_void_p
    ctx = (P * 1)(det._h)
    img = (P * 2)(P(imgs[0].data_ptr()), P(imgs[1].data_ptr()))
    kp = (P * 2)(P(kps[0].data_ptr()), P(None))
    ct = (P * 2)(P(cnt[0].data_ptr()), P(cnt[1].data_ptr()))
    rc = cef.lib().efx_detect_and_compute_batch_async(ctx, None, 1, img, 2, 240, 320, 320, kp, 2000, None, 0, 500, ct)
    assert rc == -1      # EFX_ERR_BAD_ARG


def test_no_batch_knob_is_the_same(cef, torch_mod):
    """EFX_NO_BATCH=1 (read when a context is created): the entry point as a loop of single-frame calls."""
    imgs = [synth.synth_frame(300, 400, seed=40 + i) for i in range(4)]
    d = [_dev(torch_mod, im) for im in imgs]
    a, _ = _run_batch(cef, torch_mod, d, 1000, cef.EfficientFeatures.BAD_512)
    os.environ["EFX_NO_BATCH"] = "1"
    try:
        b, _ = _run_batch(cef, torch_mod, d, 1000, cef.EfficientFeatures.BAD_512)
    finally:
        del os.environ["EFX_NO_BATCH"]
    for i in range(4):
        _same(a[i], b[i], f"frame {i}")


def _fuzz_image(rng, rows, cols):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        return synth.synth_frame(rows, cols, seed=int(rng.integers(1 << 30)), density=float(rng.choice([0.3, 1.0, 3.0])))
    if kind == 1:
        return synth.noise_frame(rows, cols, seed=int(rng.integers(1 << 30)))
    if kind == 2:
        return synth.powerlaw_frame(rows, cols, seed=int(rng.integers(1 << 30)), beta=float(rng.choice([1.0, 1.3, 1.6])), contrast=45.0)
    if kind == 3:
        return np.full((rows, cols), int(rng.integers(0, 256)), np.uint8)      # constant: a frame without corners inside a batch
    p = int(rng.integers(3, 24))
    y, x = np.mgrid[0:rows, 0:cols]
    return ((((x // p) + (y // p)) & 1) * int(rng.integers(40, 256))).astype(np.uint8)


@pytest.mark.parametrize("seed", range(int(os.environ.get("EFX_BATCH_FUZZ", "24"))))
def test_batch_fuzz_equals_single_frame_calls(cef, torch_mod, seed):
    """Seeded fuzz: random frame size (tower, tiled and row-walking pyramids), batch size, parameters, image kinds (a batch
    mixes dense, sparse and empty frames), describer; every frame of the batch == the single-frame call, bit for bit."""
    rng = np.random.default_rng(9000 + seed)
    rows = int(rng.integers(40, 1300)); cols = int(rng.integers(40, 1700))
    if seed % 6 == 5:
        rows, cols = int(rng.integers(1500, 2400)), int(rng.integers(2600, 4000))      # beyond the tower: the chain kernels
    nb = int(rng.integers(1, 9)) if seed % 6 != 5 else int(rng.integers(2, 4))
    nctx = int(rng.integers(1, 3))
    dtype = int(rng.integers(0, 4))
    kw = dict(scale_factor=float(rng.choice([1.2, 1.2, 1.35, 1.5])), nlevels=int(rng.integers(1, 9)), first_level=int(rng.integers(0, 2)),
              fast_threshold=int(rng.choice([8, 20, 40])), nonmax_radius=int(rng.choice([0, 3, 7, 15, 20])))
    if kw["first_level"] >= kw["nlevels"]:
        kw["first_level"] = 0
    nfeatures = int(rng.choice([50, 700, 4000]))
    imgs = [_fuzz_image(rng, rows, cols) for _ in range(nb)]
    d = [_dev(torch_mod, im) for im in imgs]
    describe = bool(rng.integers(0, 4))
    out, _ = _run_batch(cef, torch_mod, d, nfeatures, dtype, nctx=nctx, describe=describe, runs=int(rng.integers(1, 3)), **kw)
    for i in range(nb):
        ref = _single(cef, torch_mod, d[i], nfeatures, dtype, describe=describe, **kw)
        _same(out[i], ref[:3], f"seed {seed} {rows}x{cols} nb {nb} nctx {nctx} dtype {dtype} {kw} frame {i}")
