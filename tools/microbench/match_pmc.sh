#!/bin/bash
# usage (GPU box, repo root): tools/microbench/match_pmc.sh  -- SQ / LDS counters of the matcher kernels (40 000 x 40 000, 512 and 256 bit), two --pmc passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY --kernel-trace -d $O/pmc_match1 -o pmc -- python tools/microbench/match_bench.py > $O/pmc_match1.log 2>&1
python tools/pmc_summary.py $O/pmc_match1/pmc_results.db knn2 > $O/match_pmc.txt
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS --kernel-trace -d $O/pmc_match2 -o pmc -- python tools/microbench/match_bench.py > $O/pmc_match2.log 2>&1
python tools/pmc_summary.py $O/pmc_match2/pmc_results.db knn2 >> $O/match_pmc.txt
rm -rf $O/pmc_match1 $O/pmc_match2
cat $O/match_pmc.txt
