import os, sys; sys.path.insert(0, '.')
import numpy as np, torch, cef_loader
from oracle import matcher_oracle as MO
cef = cef_loader.load()
m = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING)
rng = np.random.default_rng(5)
nb = 64
for nt in (64, 96, 128, 160, 512):
    for nq in (128, 256, 300):
        q = (rng.random((nq, nb)) < 0.03).astype(np.uint8) * np.uint8(4)
        t = (rng.random((nt, nb)) < 0.03).astype(np.uint8) * np.uint8(4)
        dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
        i, d = m.knnMatch(dq, dt, 2); torch.cuda.synchronize()
        wi, wd = MO.knn2(q, t)
        i, d = i.cpu().numpy(), d.cpu().numpy()
        badr = np.nonzero((i != wi).any(axis=1) | (d != wd).any(axis=1))[0]
        print(f'nq {nq} nt {nt}: {len(badr)} rows differ', [(int(r), i[r].tolist(), d[r].tolist(), wi[r].tolist(), wd[r].tolist()) for r in badr[:3]])
