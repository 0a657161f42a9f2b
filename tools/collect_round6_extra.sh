#!/bin/bash
# usage (GPU box, repo root): tools/collect_round6_extra.sh   -- the round-6 artefacts beside tools/collect_round.sh r06: the bench
# lines (default command, the driver's --steps 20 --warmup 5, --force-dist), the batch sweep, kernel stats on the natural-statistics
# frames, a three-stream timeline, select_kernel's phases (an instrumented rebuild: LAST).  Everything lands in gpurun_out/r06_*.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 600 python bench.py > $O/r06_bench.json 2> $O/r06_bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_steps20.json 2>> $O/r06_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --no-configs --no-cpu-baseline > $O/r06_bench_force_dist.json 2>> $O/r06_bench.err
timeout 900 bash tools/microbench/batch_sweep.sh 2>&1 | grep -v amdgpu.ids > $O/r06_batch_sweep.txt
for dt in HASH_SIFT_512; do for e in 0 1; do
  if [ $e = 1 ]; then export EFX_NO_BATCH=1; else unset EFX_NO_BATCH; fi
  timeout 120 python tools/microbench/batch_throughput.py fhd 16 2 1.5 10000 $dt 2>&1 | grep -v amdgpu.ids | sed "s/^/EFX_NO_BATCH=$e /" >> $O/r06_batch_sweep.txt
  timeout 120 python tools/microbench/batch_throughput.py 4k 8 2 1.5 40000 $dt 2>&1 | grep -v amdgpu.ids | sed "s/^/EFX_NO_BATCH=$e /" >> $O/r06_batch_sweep.txt
done; done
unset EFX_NO_BATCH
for nf in 10000 5000; do timeout 120 python tools/microbench/batch_throughput.py fhd 16 2 1.5 $nf 2>&1 | grep -v amdgpu.ids >> $O/r06_batch_sweep.txt; done
timeout 300 tools/microbench/natural_prof.sh 1.3 1.0 > $O/r06_natural.txt 2>&1
mv $O/natural_1.3_kernel_stats.csv $O/r06_natural_1.3_kernel_stats.csv; mv $O/natural_1.0_kernel_stats.csv $O/r06_natural_1.0_kernel_stats.csv
rm -rf $O/prof_tl; rocprofv3 --kernel-trace -f csv -d $O/prof_tl -o tl -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-configs --sustain-seconds 0 > /dev/null 2>&1
python tools/timeline.py $(find $O/prof_tl -name "*kernel_trace.csv" | head -1) > $O/r06_timeline.txt 2>&1; rm -rf $O/prof_tl
(tools/microbench/sel_timing.sh 8k; tools/microbench/sel_timing.sh nat) > $O/r06_select_phases.txt 2>&1
tail -c 400 $O/r06_bench_steps20.json | head -c 0; python - <<'PY'
import json
for f in ("r06_bench.json", "r06_bench_steps20.json", "r06_bench_force_dist.json"):
    try:
        j = json.loads([l for l in open("gpurun_out/" + f) if l.startswith("{")][-1])
        print(f, j["value"], j["ms_per_step"], (j.get("sustained") or {}).get("value"), j["roofline"]["frac"], j.get("rccl_exercised"))
    except Exception as e:
        print(f, "FAILED", e)
PY
