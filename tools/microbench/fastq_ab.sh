cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_reference_table_pins.py tests/test_gpu_parity.py tests/test_arena.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
for v in 1 0 1 0; do
  make -s -C cuda-efficient-features_amd/csrc clean >/dev/null 2>&1
  make -s -C cuda-efficient-features_amd/csrc EXTRA="-DEFX_FAST_QUICK16=$v" 2>&1 | grep -E "error" | head -3
  echo "== QUICK16=$v"
  tools/microbench/prof_dbg.sh 0 sw_$v 12 | grep -E "fast_kernel|harris|nms" | cut -d, -f1,4
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_frame'])"
done
echo "== natural frame kernel stats (QUICK16=0 build)"
python tools/microbench/natural_prof.py 2>&1 | tail -15
