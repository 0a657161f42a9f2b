import sys, ctypes; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import cef_loader
from oracle import pyoracle as O
from tools import synth
from test_input_stage import colour_frame
cef = cef_loader.load()
up = cef.Uploader()
det = cef.EfficientFeatures.create(2000, dtype=cef.EfficientFeatures.BAD_256)
stream = torch.cuda.Stream()
keep = []; outs = []
for k in range(5):
    ch = (1, 3, 4, 3, 1)[k]
    src = synth.synth_frame(300, 400, seed=40 + k) if ch == 1 else colour_frame(300, 400, ch, seed=40 + k)
    h = cef.host_alloc(src.shape); h[...] = src; keep.append(h)
    d_ptr, pitch, rows, cols = up.upload(h, stream=stream)
    kps = torch.zeros((5, 2000), dtype=torch.float32, device="cuda"); desc = torch.zeros((2000, 32), dtype=torch.uint8, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = cef.lib().efx_detect_and_compute_async(det._h, ctypes.c_void_p(d_ptr), rows, cols, ctypes.c_size_t(pitch), ctypes.c_void_p(kps.data_ptr()), ctypes.c_size_t(kps.stride(0) * 4), ctypes.c_void_p(desc.data_ptr()), ctypes.c_size_t(32), 2000, ctypes.c_void_p(cnt.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
    if '--sync' in sys.argv: stream.synchronize()
    outs.append((kps, desc, cnt, src)); continue
    gray = src if src.ndim == 2 else O.bgr2gray(src)
    ref = O.detect_and_compute(gray, nfeatures=2000, desc_type=O.BAD_256)
    st = det.lastLevelStats()
    print('frame', k, 'ch', ch, 'pitch', pitch, 'n', int(cnt.item()), ref['n'], 'cand', [s['n_candidates'] for s in st], list(ref['stats']['n_candidates']))
    for level in range(3):
        got = det.copyLevel(level, rows, cols).cpu().numpy(); want = O.pyramid_level(gray, level)
        bad = np.argwhere(got != want)
        print('   level', level, 'differ', len(bad), '' if not len(bad) else ('rows %d..%d cols %d..%d' % (bad[:,0].min(), bad[:,0].max(), bad[:,1].min(), bad[:,1].max())))

stream.synchronize()
for k, (kps, desc, cnt, src) in enumerate(outs):
    gray = src if src.ndim == 2 else O.bgr2gray(src)
    ref = O.detect_and_compute(gray, nfeatures=2000, desc_type=O.BAD_256)
    n = int(cnt.item())
    same = n == ref['n'] and np.array_equal(kps[:, :n].cpu().numpy().view(np.uint32), ref['kps'].view(np.uint32))
    print('frame', k, 'channels', src.shape[2] if src.ndim == 3 else 1, 'n', n, ref['n'], 'keypoints equal', same)
