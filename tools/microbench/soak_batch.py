"""Determinism soak of the BATCHED path (the headline's form: three contexts / streams, every context's frames of a step through one
launch of every kernel): `nf` frames per step, the outputs of every step compared on the device with the single-frame results of the
same frames.  usage: soak_batch.py <seconds> [size: 8k | 4k | fhd] [frames per step: 8] [descriptor: BAD_512 | HASH_SIFT_512 ...]"""
import sys, time; sys.path.insert(0, '.')
import torch, cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
SIZE = sys.argv[2] if len(sys.argv) > 2 else '8k'
F = int(sys.argv[3]) if len(sys.argv) > 3 else 8
DT = getattr(EF, sys.argv[4] if len(sys.argv) > 4 else 'BAD_512')
ROWS, COLS = {'8k': (4320, 7680), '4k': (2160, 3840), 'fhd': (1080, 1920)}[SIZE]
NF = 40000
frames = [torch.from_numpy(synth.synth_frame(ROWS, COLS, seed=1000 + k)).cuda() for k in range(F)]
dets = [EF.create(NF, dtype=DT) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
ref = []
for f in frames:
    kps, desc, cnt = dets[0].detectAndComputeAsync(f); torch.cuda.synchronize()
    n = int(cnt.item()); ref.append((n, kps[:, :n].clone(), desc[:n].clone()))
dsz = ref[0][2].shape[1]
kps = [torch.zeros((5, NF), dtype=torch.float32, device='cuda') for _ in range(F)]
desc = [torch.zeros((NF, dsz), dtype=torch.uint8, device='cuda') for _ in range(F)]
cnt = [torch.zeros(1, dtype=torch.int32, device='cuda') for _ in range(F)]
def rot(l, r): return l[r:] + l[:r]
batches = [cef.Batch(dets, streams, rot(frames, r), rot(kps, r), rot(desc, r), rot(cnt, r), NF) for r in range(3)]
bad = torch.zeros(F, dtype=torch.int64, device='cuda')
t0 = time.time(); steps = 0
while time.time() - t0 < secs:
    for r in range(3):
        batches[r].run()
        # the checks of a step run on the default stream behind all three streams of the step; the next step waits for them
        ev = [torch.cuda.Event() for _ in streams]
        for s, e in zip(streams, ev): e.record(s)
        for e in ev: torch.cuda.current_stream().wait_event(e)
        for k in range(F):
            n, rk, rd = ref[k]
            bad[k] += (cnt[k].to(torch.int64).sum() != n).to(torch.int64) + (kps[k][:, :n] != rk).any().to(torch.int64) + (desc[k][:n] != rd).any().to(torch.int64)
        e2 = torch.cuda.Event(); e2.record(torch.cuda.current_stream())
        for s in streams: s.wait_event(e2)
        steps += 1
    torch.cuda.synchronize()
print('size', SIZE, 'frames', steps * F, 'mismatching results per frame slot', bad.cpu().tolist(), 'in', round(time.time() - t0, 1), 's')
