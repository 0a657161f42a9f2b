"""select_kernel with and without the radix passes (huge nfeatures -> no quota cut), run under rocprofv3."""
import sys; sys.path.insert(0, '.')
import torch, cef_loader
from tools import synth
cef = cef_loader.load()
img = torch.from_numpy(synth.synth_frame(4320, 7680, seed=1000)).cuda()
for nf in (40000, 400000):
    det = cef.EfficientFeatures.create(nf, dtype=1)
    kps = torch.zeros((5, nf), dtype=torch.float32, device='cuda'); cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
    for _ in range(4): det.detectAsync(img, kps, cnt)
    torch.cuda.synchronize(); print(nf, int(cnt.item()))
