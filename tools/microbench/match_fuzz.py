"""Randomised agreement of the three Hamming kernels (FP4 matrix cores, int8 matrix cores, popcount): random set sizes, 256 / 512
bit, descriptors drawn from a small pool (ties everywhere) or with few set bits.  usage (GPU box, repo root):
python tools/microbench/match_fuzz.py [cases] [seed]"""
import os, sys; sys.path.insert(0, '.')
import numpy as np, torch, cef_loader
cef = cef_loader.load()
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
def mk(env):
    for k in ('EFX_MATCH_NO_MFMA', 'EFX_MATCH_NO_FP4'): os.environ.pop(k, None)
    if env: os.environ[env] = '1'
    m = cef.BFMatcher.create(cef.BFMatcher.NORM_HAMMING)        # the knobs are read when a matcher is created
    os.environ.pop(env, None) if env else None
    return m
m_fp4, m_i8, m_pop = mk(None), mk('EFX_MATCH_NO_FP4'), mk('EFX_MATCH_NO_MFMA')
rng = np.random.default_rng(seed)
bad = 0
for c in range(ncase):
    nb = int(rng.choice([32, 64]))
    nq, nt = int(rng.integers(128, 6000)), int(rng.integers(64, 9000))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        q = rng.integers(0, 256, size=(nq, nb), dtype=np.uint8); t = rng.integers(0, 256, size=(nt, nb), dtype=np.uint8)
    elif kind == 1:                                              # a small pool: many exact duplicates
        pool = rng.integers(0, 256, size=(int(rng.integers(2, 40)), nb), dtype=np.uint8)
        q = pool[rng.integers(0, len(pool), nq)]; t = pool[rng.integers(0, len(pool), nt)]
    elif kind == 2:                                              # few set bits: distances from a handful of values
        q = (rng.random((nq, nb)) < 0.03).astype(np.uint8) * np.uint8(1 << int(rng.integers(0, 8)))
        t = (rng.random((nt, nb)) < 0.03).astype(np.uint8) * np.uint8(1 << int(rng.integers(0, 8)))
    else:                                                        # trains are noisy copies of the queries
        q = rng.integers(0, 256, size=(nq, nb), dtype=np.uint8)
        t = q[rng.integers(0, nq, nt)] ^ ((rng.random((nt, nb)) < 0.02).astype(np.uint8) * np.uint8(16))
    dq, dt = torch.from_numpy(np.ascontiguousarray(q)).cuda(), torch.from_numpy(np.ascontiguousarray(t)).cuda()
    a, b, d = m_fp4.knnMatch(dq, dt, 2), m_i8.knnMatch(dq, dt, 2), m_pop.knnMatch(dq, dt, 2)
    torch.cuda.synchronize()
    ok = torch.equal(a[0], d[0]) and torch.equal(a[1], d[1]) and torch.equal(b[0], d[0]) and torch.equal(b[1], d[1])
    if not ok:
        bad += 1
        from oracle import matcher_oracle as MO
        widx, wdist = MO.knn2(q, t)
        def cmp(x):
            i, dd = x[0].cpu().numpy(), x[1].cpu().numpy()
            return int((i != widx).any(axis=1).sum()), int((dd != wdist).any(axis=1).sum())
        print(f'case {c} MISMATCH nb {nb} nq {nq} nt {nt} kind {kind}: rows differing from the oracle (idx, dist): fp4 {cmp(a)} int8 {cmp(b)} popcount {cmp(d)}')
        if bad <= 2:
            i = a[0].cpu().numpy(); r = np.nonzero((i != widx).any(axis=1))[0][:3]
            for rr in r: print('   query', rr, 'fp4', i[rr], a[1].cpu().numpy()[rr], 'oracle', widx[rr], wdist[rr], 'pop', d[0].cpu().numpy()[rr])
print(f'{ncase} cases, {bad} mismatching')
