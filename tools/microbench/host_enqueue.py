"""Host time of one batched enqueue (8 frames x ~24 launches) vs GPU time per step (investigation helper)."""
import sys, time; sys.path.insert(0, '.')
import torch, cef_loader
from tools import synth
cef = cef_loader.load()
frames = [torch.from_numpy(synth.synth_frame(4320, 7680, seed=1000 + k)).cuda() for k in range(2)] * 4
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dets = [cef.EfficientFeatures.create(40000, dtype=1) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
kps = [torch.zeros((5, 40000), dtype=torch.float32, device='cuda') for _ in range(8)]
desc = [torch.zeros((40000, 64), dtype=torch.uint8, device='cuda') for _ in range(8)]
cnt = [torch.zeros(1, dtype=torch.int32, device='cuda') for _ in range(8)]
b = cef.Batch(dets, streams, frames, kps, desc, cnt, 40000)
for _ in range(3): b.run()
torch.cuda.synchronize()
t0 = time.perf_counter(); b.run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'streams {NS}: host enqueue of 8 frames {1e3 * (t1 - t0):.3f} ms, until GPU done {1e3 * (t2 - t0):.3f} ms')
t0 = time.perf_counter()
for _ in range(10): b.run()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'10 steps: host {1e3 * (t1 - t0) / 10:.3f} ms/step, total {1e3 * (t2 - t0) / 10:.3f} ms/step')
