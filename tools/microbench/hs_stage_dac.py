"""HashSIFT stage times inside detectAndCompute (blur path): EFX_DEBUG_HS knobs (investigation helper)."""
import os, sys; sys.path.insert(0, '.')
import torch, cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
img = torch.from_numpy(synth.synth_frame(4320, 7680, seed=1000)).cuda()
d = EF.create(40000, dtype=EF.HASH_SIFT_512)
kps = torch.zeros((5, 40000), dtype=torch.float32, device='cuda'); cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
desc = torch.zeros((40000, 64), dtype=torch.uint8, device='cuda')
for dbg in (5, 1, 2, 4, 0):
    os.environ['EFX_DEBUG_HS'] = str(dbg)
    d.detectAndComputeAsync(img, kps, desc, cnt); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): d.detectAndComputeAsync(img, kps, desc, cnt)
    b.record(); torch.cuda.synchronize()
    print('dbg', dbg, 'detectAndCompute ms', a.elapsed_time(b) / 3)
