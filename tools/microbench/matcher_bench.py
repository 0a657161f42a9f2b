import sys, time; sys.path.insert(0, '.')
import torch, cef_loader
cef = cef_loader.load()
m = cef.BFMatcher.create()
for nbytes in (32, 64):
    for n in (10000, 40000):
        q = torch.randint(0, 256, (n, nbytes), dtype=torch.uint8, device='cuda'); t = torch.randint(0, 256, (n, nbytes), dtype=torch.uint8, device='cuda')
        m.knnMatch(q, t); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): m.knnMatch(q, t)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f'knn2 {n}x{n} x {nbytes*8} bit: {ms:.3f} ms  ({n*n/ms/1e6:.1f} Gpairs/s, {n*n*nbytes*2/ms/1e9:.2f} TB/s of descriptor bytes compared)')
