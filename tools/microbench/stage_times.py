"""Times detectAsync on one 8K frame for several EFX_DEBUG stage knobs (investigation helper)."""
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
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): det.detectAsync(img, kps, cnt)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000
for name, dbg in [('full', 0), ('pyr: no fast at all', 1), ('pyr: quick-reject all', 2), ('pyr: no harris', 4), ('pyr: no resize', 8),
                  ('pyr: only load (1|8)', 9), ('nms: ret after hdr', 16), ('nms: ret after staging', 32), ('nms: ret after pass1', 48)]:
    print(f'{name:28s} {run(dbg):9.1f} us', det.lastLevelStats()[0] if dbg == 0 else '')
