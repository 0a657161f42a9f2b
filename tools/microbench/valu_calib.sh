#!/bin/bash
# usage (GPU box, repo root): tools/microbench/valu_calib.sh <tag>
# What do the SQ "VALU busy" counters read on kernels whose VALU pipe is saturated BY CONSTRUCTION (tools/microbench/valu_rate
# --calib: 8 waves per SIMD issuing one instruction class), full-rate and half-rate ones?  The same counters are then
# collected over the product's kernels (tools/pmc_run.sh valu ...), so that "fraction of the VALU pipe's time in use" is a
# measured figure per kernel, independent of the instruction mix.
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
CTR="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
rocprofv3 --pmc $CTR --kernel-trace -d $O/pmc_calib -o pmc -- tools/microbench/valu_rate --calib > $O/valu_calib.log 2>&1
{ grep "^calib" $O/valu_calib.log; python tools/pmc_summary.py $O/pmc_calib/pmc_results.db; } > $O/${tag}_valu_calib.txt
rm -rf $O/pmc_calib
tools/pmc_run.sh valu $CTR > /dev/null
python tools/pmc_summary.py $O/pmc_valu/pmc_results.db > $O/${tag}_pmc_valu.txt; rm -rf $O/pmc_valu
cat $O/${tag}_valu_calib.txt | head -80
