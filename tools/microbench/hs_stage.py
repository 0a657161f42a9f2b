"""HashSIFT compute-only stage times (C4 workload: 4K frame, 40 000 keypoints) with the EFX_DEBUG_HS stage knobs.
Needs a debug build: make -C cuda-efficient-features_amd/csrc clean all EXTRA=-DEFX_DEBUG_BUILD (investigation helper)."""
import os, sys; sys.path.insert(0, '.')
import torch, cef_loader
from tools import workloads
cef = cef_loader.load(); EF = cef.EfficientFeatures
img = torch.from_numpy(workloads.frame_c34()).cuda()
det = EF.create(workloads.N40K, 1.2, 8, 0, 20, workloads.C34_NMS_RADIUS, EF.BAD_256)
kps, cnt = det.detectAsync(img); torch.cuda.synchronize(); n = int(cnt.item())
desc = torch.zeros((n, 64), dtype=torch.uint8, device='cuda')
names = {5: 'after blur / window', 1: '+ warp patch', 6: '+ gradients, votes without the atomics', 7: 'all, orientation LUT replaced by arithmetic', 8: 'all, no weight LUT', 2: '+ gradients, votes', 3: '+ fixed -> float, fold', 4: '+ normalise', 0: 'everything (+ store, projection)'}
order = (5, 1, 6, 2, 3, 4, 0, 7, 8)
if '--only' in sys.argv: order = (int(sys.argv[sys.argv.index('--only') + 1]),)
for dbg in order:
    os.environ['EFX_DEBUG_HS'] = str(dbg)
    d = EF.create(workloads.N40K, dtype=EF.HASH_SIFT_512)          # the knob is read when the describer is created
    d.computeAsync(img, kps, n=n, descriptors=desc); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): d.computeAsync(img, kps, n=n, descriptors=desc)
    b.record(); torch.cuda.synchronize()
    print(f'n {n} dbg {dbg} {names[dbg]:34s} compute ms {a.elapsed_time(b) / 5:.4f}')
