#!/bin/bash
# INVESTIGATION (GPU box): bad_det_kernel's VALU / LDS instructions and time up to phase n (-DBAD_DET_STOP=n builds:
# 0 window loads + taps, 1 + blur row pass, 2 + blur column pass, 3 + integral rows, 4 + integral columns; none: all).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 0 1 2 3 4 full; do
  fl=""; [ $n != full ] && fl="-DBAD_DET_STOP=$n"
  (cd cuda-efficient-features_amd/csrc && rm -f bad_kernel.o hashsift_kernels.o && make -s EXTRA="$fl" 2>&1 | grep -E " error" | head -3)
  rm -rf gpurun_out/pmc_bp
  timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d gpurun_out/pmc_bp -o pmc -- python bench.py --steps 1 --warmup 1 --frames-per-step 2 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > gpurun_out/pmc_bp.log 2>&1 < /dev/null
  echo "== stop after phase $n: $(python tools/pmc_summary.py gpurun_out/pmc_bp/pmc_results.db bad_det | grep -E 'SQ_INSTS' | awk '{printf "%s %.2f M  ", $1, $NF/1e6}')"
  python - <<'P'
import sqlite3
con = sqlite3.connect("gpurun_out/pmc_bp/pmc_results.db")
r = con.execute("select avg(duration)/1e3, count(*) from kernels where name like '%bad_det%'").fetchone()
print("   avg %.1f us over %d launches (under the counter pass)" % r)
P
done
rm -rf gpurun_out/pmc_bp
