#!/bin/bash
# usage (GPU box, repo root): tools/microbench/hs_prof.sh <tag>  -- HashSIFT compute path (C4 workload): kernel stats + SQ / LDS counters
tag=${1:-hs}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o hs -- python tools/microbench/hs_stage.py --only 0 > $O/prof_$tag.log 2>&1
python tools/prof_summary.py $O/prof_$tag/hs_results.db $O/${tag}_kernel_stats.csv | head -8 | cut -c1-110; rm -rf $O/prof_$tag
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace -d $O/pmc_$tag -o pmc -- python tools/microbench/hs_stage.py --only 0 > $O/pmc_$tag.log 2>&1
python tools/pmc_summary.py $O/pmc_$tag/pmc_results.db > $O/${tag}_pmc_sq.txt; rm -rf $O/pmc_$tag
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace -d $O/pmc_$tag -o pmc -- python tools/microbench/hs_stage.py --only 0 > $O/pmc_$tag.log 2>&1
python tools/pmc_summary.py $O/pmc_$tag/pmc_results.db > $O/${tag}_pmc_lds.txt; rm -rf $O/pmc_$tag
grep -A9 "patch_sift" $O/${tag}_pmc_sq.txt | head -12; grep -A7 "patch_sift" $O/${tag}_pmc_lds.txt | head -9
