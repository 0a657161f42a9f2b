"""Times the pyramid+FAST launches (HIP events per launch) and the whole detect call on one 8K frame for several
EFX_DEBUG stage knobs (investigation helper)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import cef_loader; cef = cef_loader.load()
from tools import synth
img = torch.from_numpy(synth.synth_frame(4320, 7680, seed=1000)).cuda()
det = cef.EfficientFeatures.create(40000, dtype=1)
kps = torch.zeros((5, 40000), dtype=torch.float32, device='cuda'); cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
def run(dbg, reps=5):
    os.environ['EFX_DEBUG'] = str(dbg)
    det.detectAsync(img, kps, cnt); torch.cuda.synchronize()
    det.profileEnable(reps * 8)
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): det.detectAsync(img, kps, cnt)
    b.record(); torch.cuda.synchronize()
    ms, lvl = det.profileRead()
    det.profileEnable(0)
    fast = ms[lvl == 0].mean() * 1000
    chain = ms[lvl >= 100].sum() * 1000 / reps
    return a.elapsed_time(b) / reps * 1000, fast, chain
cases = [('full', 0), ('fast: quick-reject all', 2), ('fast: no harris', 4), ('nms: ret after hdr', 16), ('nms: ret after staging', 32), ('nms: ret after pass1', 48)]
if len(sys.argv) > 1: cases = [(f'dbg {v}', int(v)) for v in sys.argv[1:]]
for name, dbg in cases:
    tot, fast, chain = run(dbg)
    print(f'{name:26s} detect {tot:8.1f} us | fast_kernel {fast:7.1f} us | resize chain {chain:6.1f} us', det.lastLevelStats()[:1])
