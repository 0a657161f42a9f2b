#!/bin/bash
# INVESTIGATION (GPU box): the kernel timeline of ONE detectAndCompute BAD512 call (8K; TRACE_ROWS / TRACE_COLS for other sizes),
# call-then-wait protocol: start / end of every kernel relative to the call's first.  ($1: value of EFX_OVERLAP for a build
# with tools/experiments/overlap_chain_level0.patch applied; ignored by the shipped library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/oct
cat > /tmp/oct.py <<'P'
import sys; sys.path.insert(0, '.')
import torch, cef_loader
from tools import synth
cef = cef_loader.load(); EF = cef.EfficientFeatures
import os
R, C = int(os.environ.get('TRACE_ROWS', 4320)), int(os.environ.get('TRACE_COLS', 7680))
img = torch.from_numpy(synth.synth_frame(R, C, seed=1000)).cuda()
d = EF.create(40000, dtype=EF.BAD_512)
k, desc, cnt = d.detectAndComputeAsync(img); torch.cuda.synchronize()
for _ in range(6):
    d.detectAndComputeAsync(img, keypoints=k, descriptors=desc, count=cnt); torch.cuda.synchronize()
P
EFX_OVERLAP=$1 timeout 200 rocprofv3 --kernel-trace -d gpurun_out/oct -o t -- python /tmp/oct.py > gpurun_out/oct.log 2>&1 < /dev/null
python - <<'P'
import sqlite3, glob
con = sqlite3.connect(glob.glob("gpurun_out/oct/**/t_results.db", recursive=True)[0])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = con.execute("select start, end, name, %s from kernels order by start" % qcol).fetchall()
rows = [r for r in rows if "at::" not in r[2] and "rocclr" not in r[2]]
# the last call: everything behind the last gap of more than 50 us between two kernels
i0 = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(r[1] for r in rows[max(0, i - 8):i]) > 50000: i0 = i
t0 = min(r[0] for r in rows[i0:])
for r in rows[i0:]:
    nm = r[2].split("::")[-1].split("(")[0][:28]
    print("%8.1f %8.1f  %6.1f us  q%s  %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], nm))
print("call: %.1f us" % ((max(r[1] for r in rows[i0:]) - t0) / 1e3))
P
rm -rf gpurun_out/oct
