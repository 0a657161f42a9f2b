import sys; sys.path.insert(0,'.')
import numpy as np, torch
import cef_loader; cef=cef_loader.load()
from oracle import pyoracle as O
from tools import synth
img = synth.synth_frame(480, 640, seed=4)
kps = synth.random_keypoints(480, 640, 2000, seed=17)
hs = cef.HashSIFT.create(1.0, cef.HashSIFT.SIZE_256_BITS)
d_img=torch.from_numpy(img).cuda(); d_k=torch.from_numpy(kps).cuda()
resp,T = hs.debug(d_img,d_k,max_size=31.0); torch.cuda.synchronize()
resp=resp.cpu().numpy(); T=T.cpu().numpy()
want=O.hashsift_responses(img,kps); wT,wd=O.hashsift_project(want,256)
d=np.abs(resp-want)
print('frac elems differ', (d>0).mean(), 'kp with any diff', (d>0).any(1).sum(), 'of', len(kps))
print('hist', np.bincount(d.astype(int).ravel()))
bad=np.argsort(-d.max(1))[:10]
for b in bad: print(b, kps[b], d[b].max(), (d[b]>0).sum())
same=(d==0).all(1)
print('T maxdiff on same', np.abs(T[same]-wT[same]).max(), 'T abs max', np.abs(wT).max(), 'median', np.median(np.abs(wT)))
# integer-position-only & special angles
for name,mask in [('int pos', np.arange(len(kps))<len(kps)//2), ('angle -1', kps[:,3]==-1), ('angle 0', kps[:,3]==0)]:
    print(name, (d[mask]>0).any(1).mean())
