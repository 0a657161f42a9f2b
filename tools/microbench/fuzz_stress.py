"""Repeat one detector fuzz case many times (run several copies at once to share the GPU, tools/microbench/stress16.sh):
looks for timing-dependent results.
  default          a new context and a new upload of the frame every repetition (the mode that shows ~1 wrong frame in 2500
                   when 16 copies share an MI355X)
  --persist        one context, one device image for all repetitions (0 wrong frames in 96 000)
  --persist-ctx / --persist-img   only one of the two
  --alternate      two different frames take turns (what a lost store leaves behind differs from what belongs there)
  --churn / --churn-run           additionally create (and run) a throw-away context every repetition
With a debug build of the library (make EXTRA=-DEFX_DEBUG_BUILD) a mismatching frame's harris / nms stages are repeated on
the frame's buffers and digests of the corner arena printed; EFX_TRACE=1 then traces every launch."""
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
want_c = list(ref['stats']['n_candidates'])
bad = 0
# --alternate: two different frames take turns (what a lost store leaves behind then differs from what belongs there)
img_b = np.ascontiguousarray(img[::-1, ::-1])
ref_b = O.detect_and_compute(img_b, desc_type=desc_type, mask=None if mask is None else np.ascontiguousarray(mask[::-1, ::-1]), **kw)
alt = '--alternate' in sys.argv and mask is None
d_img_b = torch.from_numpy(img_b).cuda()
persist = '--persist' in sys.argv or '--persist-ctx' in sys.argv
persist_img = '--persist' in sys.argv or '--persist-img' in sys.argv
det = None
d_img0 = torch.from_numpy(img).cuda()
d_mask = None if mask is None else torch.from_numpy(mask).cuda()
for rep in range(reps):
    if det is None or not persist:
        det = cef.EfficientFeatures.create(kw["nfeatures"], kw["scale_factor"], kw["nlevels"], kw["first_level"], kw["fast_threshold"], kw["nonmax_radius"], max(desc_type, 0))
    if '--churn' in sys.argv:
        tmp = cef.EfficientFeatures.create(100, dtype=1); del tmp          # an unused context: hipMalloc + hipMemcpy + hipFree
    if '--churn-run' in sys.argv:
        tmp = cef.EfficientFeatures.create(100, dtype=1); tmp.detectAndComputeAsync(d_img0); torch.cuda.synchronize(); del tmp
    d_img = d_img0 if persist_img else torch.from_numpy(img).cuda()
    use_b = alt and (rep & 1)
    if use_b: d_img = d_img_b
    want = list((ref_b if use_b else ref)['stats']['n_after_nms']); want_c = list((ref_b if use_b else ref)['stats']['n_candidates'])
    kps, desc, cnt = det.detectAndComputeAsync(d_img, mask=d_mask) if d_mask is not None else det.detectAndComputeAsync(d_img)
    torch.cuda.synchronize()
    if (rep == 0 or '--always' in sys.argv) and hasattr(cef.lib(), 'efx_debug_rerun'):
        import ctypes
        tot = (ctypes.c_int * 8)()
        cef.lib().efx_debug_rerun.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        cef.lib().efx_debug_rerun(det._h, 2, tot, 8)
        if rep == 0: print('   first frame: rerun nms survivors', list(tot)[:4], 'digests xy, corners, cmax, pyramid+hdr', list(tot)[4:])
        elif list(tot)[:3] != want[:3]: print('   ALWAYS-rerun mismatch at rep', rep, list(tot))
    ls = det.lastLevelStats()
    st = [s['n_after_nms'] for s in ls]
    stc = [s['n_candidates'] for s in ls]
    if st != want or stc != want_c:
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
        import ctypes
        for stages in ((2, 2, 3, 3) if hasattr(cef.lib(), 'efx_debug_rerun') else ()):
            tot = (ctypes.c_int * 8)()
            cef.lib().efx_debug_rerun.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
            rc = cef.lib().efx_debug_rerun(det._h, stages, tot, 8)
            print('   rerun stages', stages, 'rc', rc, 'survivors', list(tot)[:4], 'want', want[:4], 'digests', list(tot)[4:])
        if bad <= 5: print('rep', rep, 'cand gpu', stc[:4], 'want', want_c[:4], 'nms gpu', st, 'want', want, 'count', int(cnt.item()), 'lastCount', det.lastCount())
print('seed', seed, 'reps', reps, 'mismatches', bad)
