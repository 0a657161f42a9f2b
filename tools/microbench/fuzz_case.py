#!/usr/bin/env python3
"""One fuzz case in detail (tests/test_gpu_fuzz.py): python tools/microbench/fuzz_case.py <seed>
Compares pyramid levels, keypoints and descriptor bytes with the oracle and says where the HashSIFT bytes differ."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import cef_loader
from oracle import pyoracle
from tests.test_gpu_fuzz import _case
seed = int(sys.argv[1])
cef = cef_loader.load(); EF = cef.EfficientFeatures
img, mask, dt, kw = _case(seed)
print("seed", seed, img.shape, "desc_type", dt, "mask", mask is not None, kw)
ref = pyoracle.detect_and_compute(img, desc_type=dt, mask=mask, **kw)
det = EF.create(kw["nfeatures"], kw["scale_factor"], kw["nlevels"], kw["first_level"], kw["fast_threshold"], kw["nonmax_radius"], dt)
d_img = torch.from_numpy(np.ascontiguousarray(img)).cuda()
d_mask = torch.from_numpy(np.ascontiguousarray(mask)).cuda() if mask is not None else None
kps, desc, cnt = det.detectAndComputeAsync(d_img, mask=d_mask)
torch.cuda.synchronize()
n = int(cnt.item())
print("n", n, ref["n"])
for level in range(kw["nlevels"]):
    try:
        got = det.copyLevel(level, img.shape[0], img.shape[1]).cpu().numpy()
        want = pyoracle.pyramid_level(img, level, scale_factor=kw["scale_factor"])
        print("level", level, got.shape, "pixels differing:", int(np.count_nonzero(got != want)))
    except Exception as e:
        print("level", level, "error", e)
g = kps[:, :n].cpu().numpy().view(np.uint32); r = ref["kps"].view(np.uint32)
print("keypoint rows equal:", bool(np.array_equal(g, r)))
d = desc[:n].cpu().numpy()
bad = np.argwhere(d != ref["desc"])
print("descriptor bytes differing:", len(bad), "of", d.size)
kk = sorted(set(int(b[0]) for b in bad))
print("keypoints affected:", len(kk))
for k in kk[:12]:
    col = ref["kps"][:, k]
    print("  kp", k, "xy", col[0:1].view(np.int16) if False else (int(ref["kps"].view(np.uint32)[0, k] & 0xffff), int(ref["kps"].view(np.uint32)[0, k] >> 16)), "octave", col[3].view(np.int32) if hasattr(col[3], "view") else col[3], "bytes", [int(b[1]) for b in bad if b[0] == k][:8])
