#!/bin/bash
# usage (GPU box): tools/microbench/stress16.sh <seed> <reps> [nproc]  -- N copies of fuzz_stress.py sharing the GPU
seed=${1:-233217}; reps=${2:-1500}; np=${3:-16}
for i in $(seq 1 $np); do timeout 600 python tools/microbench/fuzz_stress.py $seed $reps $EXTRA > gpurun_out/stress_$i.log 2>&1 & done
wait
grep -h "mismatches" gpurun_out/stress_*.log | awk '{s+=$6; n+=$4} END{print "frames", n, "mismatches", s}'
grep -h "rep \|rerun\|first frame\|ALWAYS\|extra\|missing\|common" gpurun_out/stress_*.log | head -${HEAD:-12} | cut -c1-300
rm -f gpurun_out/stress_*.log
