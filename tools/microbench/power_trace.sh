#!/bin/bash
# usage (GPU box, repo root): tools/microbench/power_trace.sh [bench args]  -- samples rocm-smi (sclk, socket power, temperature) every
# 0.25 s while bench.py runs; prints the samples taken during the timed region (the busiest second)
O=gpurun_out; mkdir -p $O
( while true; do echo "t $(date +%s.%N) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|Socket Graphics|Temperature \(Sensor junction\)' | sed -E 's/.*: //' | tr '\n' ' ')"; sleep 0.25; done ) > $O/power_trace.txt &
S=$!
python bench.py --no-cpu-baseline --no-configs --sustain-seconds 0 "$@" > $O/power_bench.json 2>/dev/null
kill $S
python - <<'PY'
import re
rows=[l.split() for l in open("gpurun_out/power_trace.txt") if l.startswith("t ")]
busy=[r for r in rows if any(x.replace(".","").isdigit() and float(x) > 400 for x in r[2:] if re.match(r"^[\d.]+$", x))]
print("samples", len(rows), "with power > 400 W:", len(busy))
for r in rows[::2][:60]: print(" ".join(r[2:]))
PY
tail -c 200 $O/power_bench.json
