import sys, time, gc
sys.path.insert(0, ".")
import torch, cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
img = torch.from_numpy(synth.synth_frame(4320, 7680, seed=1000)).cuda()
kps = torch.zeros((5, 40000), dtype=torch.float32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
desc = torch.zeros((40000, 64), dtype=torch.uint8, device="cuda")
def lat(tag, n=60):
    det = EF.create(40000, dtype=EF.BAD_512)
    for _ in range(5):
        det.detectAndComputeAsync(img, kps, desc, cnt); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); det.detectAndComputeAsync(img, kps, desc, cnt); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); print(f"{tag}: mean {sum(ts)/len(ts)*1e3:.4f} median {ts[len(ts)//2]*1e3:.4f} min {ts[0]*1e3:.4f}", flush=True)
lat("fresh process")
frames = [torch.from_numpy(synth.synth_frame(4320, 7680, seed=1000 + k)).cuda() for k in range(8)]
dets = [EF.create(40000, dtype=EF.BAD_512) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
K = [torch.zeros((5, 40000), dtype=torch.float32, device="cuda") for _ in range(8)]
D = [torch.zeros((40000, 64), dtype=torch.uint8, device="cuda") for _ in range(8)]
C = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(8)]
b = cef.Batch(dets, streams, frames, K, D, C, 40000)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    for _ in range(20): b.run()
    torch.cuda.synchronize()
lat("after 3 s of three-stream load (its contexts and streams alive)")
time.sleep(2.0)
lat("... 2 s idle later")
del b, dets, streams; gc.collect(); torch.cuda.synchronize()
lat("contexts and streams deleted")
