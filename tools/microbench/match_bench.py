"""Brute-force Hamming matcher: 40 000 x 40 000 descriptors, knnMatch(k = 2), ms per call (512 and 256 bit)."""
import sys; sys.path.insert(0, '.')
import os
import numpy as np, torch, cef_loader
cef = cef_loader.load()
rng = np.random.default_rng(1)
for nbytes in (64, 32):
    q = torch.from_numpy(rng.integers(0, 256, size=(40000, nbytes), dtype=np.uint8)).cuda()
    t = torch.from_numpy(rng.integers(0, 256, size=(40000, nbytes), dtype=np.uint8)).cuda()
    m = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING)
    m.knnMatch(q, t, 2); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): m.knnMatch(q, t, 2)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f'{nbytes * 8} bit: {ms:.3f} ms per 40k x 40k knnMatch = {40000 * 40000 / ms / 1e6:.0f} G pairs/s = {40000 * 40000 * nbytes * 8 * 2 / ms / 1e9:.0f} int8 TOP/s')
