"""Reproduce one detector fuzz case (tests/test_gpu_fuzz.py seed) with per-level diagnostics."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import cef_loader
from oracle import pyoracle as O
import test_gpu_fuzz as F
cef = cef_loader.load()
seed = int(sys.argv[1])
img, mask, desc_type, kw = F._case(seed)
print(img.shape, desc_type, mask is not None, kw)
for rep in range(3):
    det = cef.EfficientFeatures.create(kw["nfeatures"], kw["scale_factor"], kw["nlevels"], kw["first_level"], kw["fast_threshold"], kw["nonmax_radius"], max(desc_type, 0))
    d_img = torch.from_numpy(img).cuda()
    kps, cnt = det.detectAsync(d_img)
    torch.cuda.synchronize()
    ref = O.detect_and_compute(img, desc_type=-1, mask=None, **kw)
    st = det.lastLevelStats()
    print('rep', rep, 'gpu cand', [s['n_candidates'] for s in st], 'oracle', list(ref['stats']['n_candidates']))
    for level in range(kw["nlevels"]):
        got = det.copyLevel(level, img.shape[0], img.shape[1]).cpu().numpy()
        want = O.pyramid_level(img, level, scale_factor=kw["scale_factor"]) if 'scale_factor' in O.pyramid_level.__code__.co_varnames else O.pyramid_level(img, level)
        bad = np.argwhere(got != want) if got.shape == want.shape else None
        print('  level', level, got.shape, want.shape, 'differ', None if bad is None else len(bad), '' if bad is None or not len(bad) else ('rows %d..%d cols %d..%d' % (bad[:,0].min(), bad[:,0].max(), bad[:,1].min(), bad[:,1].max())))
