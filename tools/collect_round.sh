#!/bin/bash
# usage (GPU box, repo root): tools/collect_round.sh <tag>   e.g. r02
# Collects everything profiles/ holds for one build into gpurun_out/<tag>_*: kernel stats (rocprofv3 --kernel-trace --stats,
# one stream and the default three), SQ / LDS counters and HBM traffic (separate --pmc passes), the issue-rate and
# workgroup-rate micro-benchmarks, <tag>_counters.json (what bench.py's roofline.valu / lds / traffic read), the bench.py
# line, the BASELINE.json configurations, HashSIFT / matcher kernel stats.  Copy the files into profiles/ afterwards.
# BEFORE the GPU call, in the build container: python tools/valu_mix.py profiles/<tag>_valu_rate.txt profiles/<tag>_valu_mix.json (static
# instruction mix of the compiled kernels; counters_json.py folds it in).  AFTER it: copy gpurun_out/<tag>_* into profiles/ and rerun
# tools/counters_json.py there if the mix file changed.
# usage: tools/collect_round.sh <tag> [git commit of the tree]   (the box has no .git: pass `git rev-parse HEAD` from the build container)
tag=${1:-rXX}
export EFX_GIT_HEAD=${2:-${EFX_GIT_HEAD:-unknown}}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
# (EFX_NO_BATCH=1: one launch chain per frame, so that a kernel's average is its time for ONE frame, as in the earlier rounds' files;
# the three-stream run below is the default form: every context's frames of a step through one launch of every kernel)
EFX_NO_BATCH=1 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 --streams 1 > $O/bench_$tag.log 2>&1
python tools/prof_summary.py $O/prof_$tag/bench_results.db $O/${tag}_kernel_stats.csv > /dev/null; rm -rf $O/prof_$tag
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_3s -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs --sustain-seconds 0 > $O/bench_${tag}_3s.log 2>&1
python tools/prof_summary.py $O/prof_${tag}_3s/bench_results.db $O/${tag}_kernel_stats_3streams.csv > /dev/null; rm -rf $O/prof_${tag}_3s
tools/pmc_run.sh sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS > /dev/null
python tools/pmc_summary.py $O/pmc_sq/pmc_results.db > $O/${tag}_pmc_sq.txt; rm -rf $O/pmc_sq
tools/pmc_run.sh lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS > /dev/null
python tools/pmc_summary.py $O/pmc_lds/pmc_results.db > $O/${tag}_pmc_lds.txt; rm -rf $O/pmc_lds
tools/pmc_run.sh fetch FETCH_SIZE > /dev/null
tools/pmc_run.sh write WRITE_SIZE > /dev/null
python tools/traffic_json.py $O/pmc_fetch/pmc_results.db $O/pmc_write/pmc_results.db $O/${tag}_traffic.json > /dev/null; rm -rf $O/pmc_fetch $O/pmc_write
# (the issue-rate tables are a property of the chip: without the compiled micro-benchmarks -- tools/microbench/valu_calib.sh builds them --
# the round's committed table is used)
if [ -x tools/microbench/valu_rate ]; then tools/microbench/valu_rate > $O/${tag}_valu_rate.txt 2>&1; else cp profiles/${tag}_valu_rate.txt $O/${tag}_valu_rate.txt; fi
if [ -x tools/microbench/wg_rate ]; then tools/microbench/wg_rate > $O/wg_rate.txt 2>&1; fi
python tools/counters_json.py $O/${tag}_pmc_sq.txt $O/${tag}_pmc_lds.txt $O/${tag}_traffic.json $O/${tag}_valu_rate.txt $O/${tag}_counters.json profiles/${tag}_valu_mix.json
if [ -n "$FULL" ]; then
  python tools/bench_configs.py --out $O/${tag}_configs.json > /dev/null
  rocprofv3 --kernel-trace --stats -d $O/prof_hs -o hs -- python tools/microbench/hs_stage.py > $O/prof_hs.log 2>&1
  python tools/prof_summary.py $O/prof_hs/hs_results.db $O/${tag}_hashsift_kernel_stats.csv > /dev/null; rm -rf $O/prof_hs
  python tools/bench_input_stage.py --out $O/${tag}_input_stage.json > /dev/null 2>&1
fi
head -12 $O/${tag}_kernel_stats.csv | cut -c1-100
