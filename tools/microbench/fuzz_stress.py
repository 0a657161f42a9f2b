"""Repeat one detector fuzz case many times (run several copies at once to share the GPU): looks for timing-dependent results."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import cef_loader
from oracle import pyoracle as O
import test_gpu_fuzz as F
cef = cef_loader.load()
seed, reps = int(sys.argv[1]), int(sys.argv[2])
img, mask, desc_type, kw = F._case(seed)
ref = O.detect_and_compute(img, desc_type=desc_type, mask=mask, **kw)
want = list(ref['stats']['n_after_nms'])
bad = 0
for rep in range(reps):
    det = cef.EfficientFeatures.create(kw["nfeatures"], kw["scale_factor"], kw["nlevels"], kw["first_level"], kw["fast_threshold"], kw["nonmax_radius"], max(desc_type, 0))
    d_img = torch.from_numpy(img).cuda()
    kps, desc, cnt = det.detectAndComputeAsync(d_img)
    torch.cuda.synchronize()
    st = [s['n_after_nms'] for s in det.lastLevelStats()]
    if st != want:
        bad += 1
        if bad <= 3:
            n = int(cnt.item())
            g = kps[:, :n].cpu().numpy(); r = ref['kps']
            def keyset(a):
                loc = a[0].view(np.uint32); return {(int(l & 0xffff), int(l >> 16), int(o)): float(rs) for l, o, rs in zip(loc, a[3].view(np.int32), a[1])}
            G, R = keyset(g), keyset(r)
            print('   extra on gpu (x, y, octave): response', sorted((k, G[k]) for k in G.keys() - R.keys())[:12])
            print('   missing on gpu', sorted((k, R[k]) for k in R.keys() - G.keys())[:12])
            # responses of common keypoints
            diff = [(k, G[k], R[k]) for k in G.keys() & R.keys() if G[k] != R[k]]
            print('   common keypoints with different response', diff[:6])
        if bad <= 5: print('rep', rep, 'gpu', st, 'want', want, 'count', int(cnt.item()), 'lastCount', det.lastCount())
print('seed', seed, 'reps', reps, 'mismatches', bad)
