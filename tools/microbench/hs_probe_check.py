import sys; sys.path.insert(0,'.')
import numpy as np, torch
import cef_loader; cef=cef_loader.load()
from oracle import pyoracle as O
g=np.load('tests/golden/descriptors_probe.npz')
img,kps=g['image'],g['keypoints']
for nbits,enum in ((256,cef.HashSIFT.SIZE_256_BITS),(512,cef.HashSIFT.SIZE_512_BITS)):
    hs=cef.HashSIFT.create(1.0,enum)
    got=hs.compute(img,kps)
    print(nbits,'bytes differing',np.count_nonzero(got!=g[f'hashsift{nbits}']),'of',got.size)
    resp,T=hs.debug(torch.from_numpy(img).cuda(), torch.from_numpy(kps).cuda(), max_size=31.0)
    torch.cuda.synchronize()
    want=O.hashsift_responses(img,kps)
    d=np.abs(resp.cpu().numpy()-want)
    print(' vec elements differing',(d>0).sum(),'max',d.max(), 'rows', np.nonzero((d>0).any(1))[0][:10])
