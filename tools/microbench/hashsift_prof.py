import sys; sys.path.insert(0, '.')
import torch, cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
img = torch.from_numpy(synth.synth_frame(4320, 7680, seed=1000)).cuda()
d = EF.create(40000, dtype=EF.HASH_SIFT_512)
kps = torch.zeros((5, 40000), dtype=torch.float32, device='cuda'); cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
desc = torch.zeros((40000, 64), dtype=torch.uint8, device='cuda')
d.detectAsync(img, kps, cnt); torch.cuda.synchronize(); n = int(cnt.item())
for _ in range(3):
    d.computeAsync(img, kps, n=n, descriptors=desc); d.detectAndComputeAsync(img, kps, desc, cnt)
torch.cuda.synchronize()
