#!/bin/bash
# NMS iteration (GPU box): detector parity + fuzz, then nms_kernel's time in the density cases (order of the launches =
# order of the cases, 11 launches each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_mask_provided.py tests/test_gpu_fuzz.py -m gpu -q -x -k "detect or fuzz or golden or mask or equal_responses or extremes or cap" 2>&1 | tail -2
rocprofv3 --kernel-trace -d $O/prof_nms -o nms -- python tools/microbench/nms_density.py > $O/nms_density.log 2>&1
grep detect $O/nms_density.log
python - <<'P'
import sqlite3
con = sqlite3.connect("gpurun_out/prof_nms/nms_results.db")
r = [x[0] for x in con.execute("select duration/1e3 from kernels where name like '%nms_kernel%' order by start").fetchall()]
for i, lab in enumerate(("4k_c34_radius5", "4k_3x_density", "4k_default", "8k_default")):
    seg = r[11 * i + 1: 11 * i + 11]
    print("nms_kernel %-16s %.1f us" % (lab, sum(seg) / max(len(seg), 1)))
P
rm -rf $O/prof_nms
