#!/bin/bash
# rocprofv3 kernel stats of a batched run: tools/microbench/batch_prof.sh <size> <B> <nctx> <tag>
size=${1:-fhd}; B=${2:-16}; nctx=${3:-1}; tag=${4:-batch}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o b -- python tools/microbench/batch_throughput.py $size $B $nctx 0.5 > gpurun_out/prof_$tag.log 2>&1
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" $B <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
B = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':60s} {'calls':>6s} {'avg us':>9s} {'us/frame':>9s} {'%':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f"{r['Name'][:60]:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} {float(r['AverageNs'])/1e3/B:9.2f} {100*float(r['TotalDurationNs'])/tot:6.1f}")
PY
