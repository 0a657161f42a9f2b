#!/bin/bash
# usage (GPU box, repo root): tools/collect_round.sh <tag>   e.g. r01_h
# Collects everything profiles/ holds for one build into gpurun_out/: kernel stats (rocprofv3 --kernel-trace --stats),
# the bench.py line, the BASELINE.json configurations, SQ counters, HBM traffic (separate --pmc passes), and the
# HashSIFT / matcher kernel stats.  Copy the files you want judged into profiles/ afterwards.
tag=${1:-rXX}
tools/gpu_check.sh $tag > /dev/null
python tools/prof_summary.py gpurun_out/prof_$tag/bench_results.db gpurun_out/${tag}_kernel_stats.csv > /dev/null
python bench.py | tail -1 > gpurun_out/${tag}_bench.json
# the same default command (three frames in flight) under rocprofv3: the kernel durations stretch with the concurrency,
# these are the ones bench.py's live `roofline.avg_launch_ms` must agree with
(cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_3s -o bench -- python bench.py --no-cpu-baseline > gpurun_out/bench_${tag}_3s.log 2>&1)
python tools/prof_summary.py gpurun_out/prof_${tag}_3s/bench_results.db gpurun_out/${tag}_kernel_stats_3streams.csv > /dev/null
python tools/bench_configs.py --out gpurun_out/${tag%_*}_configs.json > /dev/null
tools/pmc_run.sh sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS > /dev/null
python tools/pmc_summary.py gpurun_out/pmc_sq/pmc_results.db > gpurun_out/${tag}_pmc_sq.txt
tools/pmc_run.sh fetch FETCH_SIZE > /dev/null
tools/pmc_run.sh write WRITE_SIZE > /dev/null
python tools/traffic_json.py gpurun_out/pmc_fetch/pmc_results.db gpurun_out/pmc_write/pmc_results.db gpurun_out/traffic.json > /dev/null
tools/microbench/prof_other.sh > /dev/null 2>&1
python tools/bench_input_stage.py --out gpurun_out/${tag%_*}_input_stage.json > /dev/null 2>&1
head -12 gpurun_out/${tag}_kernel_stats.csv | cut -c1-100
cut -c1-260 gpurun_out/${tag}_bench.json
