"""Investigate HashSIFT fuzz mismatches: for each differing bit print the oracle's pre-threshold |T| (spec S8 tolerance)."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import cef_loader
from oracle import pyoracle as O
import test_gpu_fuzz as F
cef = cef_loader.load()
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(10_000 + seed)
    rows, cols = int(rng.integers(20, 300)), int(rng.integers(20, 400))
    kindimg = int(rng.integers(0, 5))
    rng2 = np.random.default_rng(10_000 + seed); rows, cols = int(rng2.integers(20, 300)), int(rng2.integers(20, 400))
    img = F._image(rng2, rows, cols, int(rng2.integers(0, 5)))
    rng = rng2
    n = int(rng.integers(1, 300))
    kps = np.zeros((n, 4), np.float32)
    kps[:, 0] = rng.uniform(-20, cols + 20, n); kps[:, 1] = rng.uniform(-20, rows + 20, n)
    kps[:, 2] = rng.choice([31.0, 31.0, 7.0, 12.5, 48.0, 64.0, 90.0], n); kps[:, 3] = rng.uniform(0, 360, n)
    sel = rng.random(n); kps[sel < 0.1, 3] = -1.0; kps[(sel >= 0.1) & (sel < 0.15), 3] = -7.0
    if rng.random() < 0.5: kps[:, :2] = np.floor(kps[:, :2])
    scale = float(rng.choice([1.0, 1.0, 0.75, 1.5, 2.0]))
    kps[:, 2] = np.minimum(kps[:, 2], np.float32(110.0 / scale))
    kind = int(rng.integers(0, 4))
    nbits = (256, 512)[kind - 2]
    enum = (cef.HashSIFT.SIZE_256_BITS, cef.HashSIFT.SIZE_512_BITS)[kind - 2]
    got = cef.HashSIFT.create(scale, enum).compute(img, kps)
    resp = O.hashsift_responses(img, kps, scale)
    T, want = O.hashsift_project(resp, nbits)
    gb = np.unpackbits(got, axis=1); wb = np.unpackbits(want, axis=1)
    bad = np.argwhere(gb != wb)
    print(f'seed {seed}: image kind {kindimg} {img.shape} n {n} scale {scale} bits {nbits}: {len(bad)} bits differ in {len(set(bad[:,0]))} keypoints')
    tt = np.abs(np.asarray(T)[bad[:, 0], bad[:, 1]])
    print('  |T| of the differing bits: max %.3g  median %.3g ;  rows: %s' % (tt.max() if len(tt) else 0, np.median(tt) if len(tt) else 0, sorted(set(bad[:, 0]))[:10]))
    for r in sorted(set(bad[:, 0]))[:4]:
        print('  keypoint', r, kps[r], 'resp norm', float(np.abs(resp[r]).sum()), 'nonzero', int(np.count_nonzero(resp[r])))
    import torch
    hs = cef.HashSIFT.create(scale, enum)
    r2, T2 = hs.debug(torch.from_numpy(img).cuda(), torch.from_numpy(kps).cuda(), max_size=float(kps[:, 2].max()))
    torch.cuda.synchronize()
    r2 = r2.cpu().numpy()
    fx = O.hashsift_responses_fixedpoint(img, kps, scale) if 'crop_scale' in O.hashsift_responses_fixedpoint.__code__.co_varnames else None
    for r in sorted(set(bad[:, 0]))[:4]:
        d = r2[r] - resp[r]
        print('   vec diff: max abs', np.abs(d).max(), 'count', np.count_nonzero(d), 'at', np.nonzero(d)[0][:12], 'values', d[np.nonzero(d)[0][:6]])
        if fx is not None: print('   vs fixed-point model: count', np.count_nonzero(r2[r] - fx[r]))
