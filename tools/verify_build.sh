#!/bin/bash
# usage (GPU box, repo root): tools/verify_build.sh [fuzz cases per leg, default 12000] [first seed, default 900000]
# The verification legs of a final build in one call: determinism soaks (single-frame calls on three streams; the batched path the headline takes), seeded fuzz on FRESH seeds (mixed
# cases; the rows kernel forced onto small frames under four level splits; the tiled chain; harris_packed_kernel forced; batches of
# frames against single-frame calls), the three Hamming kernels against each other, sixteen processes sharing the GPU.  Prints one line per leg; gpurun_out/verify.log holds the details.
n=${1:-12000}; first=${2:-900000}
cd "$GRAFT_REPO_ROOT"; L=gpurun_out/verify.log; : > $L
echo "soak: $(timeout 300 python tools/microbench/soak.py 60 2>&1 | tail -1)" | tee -a $L
echo "soak, batched path: $(timeout 300 python tools/microbench/soak_batch.py 45 8k 8 BAD_512 2>&1 | tail -1)" | tee -a $L
echo "soak, batched path: $(timeout 300 python tools/microbench/soak_batch.py 30 fhd 16 HASH_SIFT_512 2>&1 | tail -1)" | tee -a $L
echo "fuzz mixed: $(EFX_FUZZ_CASES=$n EFX_FUZZ_FIRST=$first timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -1)" | tee -a $L
k=0
for split in "2,2,3" "1,1,1,1,1,1,1" "3,4" "4,3"; do
  k=$((k + 1))
  echo "fuzz rows kernel, split $split: $(EFX_NO_TOWER=1 EFX_ROWS_SPLIT=$split EFX_FUZZ_CASES=$((n / 4)) EFX_FUZZ_FIRST=$((first + 100000 * k)) timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k detect_and_compute 2>&1 | tail -1)" | tee -a $L
done
echo "fuzz tiled chain: $(EFX_NO_RESIZE_ROWS=1 EFX_NO_TOWER=1 EFX_FUZZ_CASES=$((n / 4)) EFX_FUZZ_FIRST=$((first + 500000)) timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k detect_and_compute 2>&1 | tail -1)" | tee -a $L
echo "fuzz packed harris kernel: $(EFX_PACK=1 EFX_FUZZ_CASES=$((n / 4)) EFX_FUZZ_FIRST=$((first + 600000)) timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k detect_and_compute 2>&1 | tail -1)" | tee -a $L
echo "fuzz frame-batched launches: $(EFX_BATCH_FUZZ=$((n / 40)) timeout 1800 python -m pytest tests/test_batch.py -m gpu -q -x -k fuzz 2>&1 | tail -1)" | tee -a $L
echo "matcher fuzz: $(timeout 900 python tools/microbench/match_fuzz.py 600 $first 2>&1 | tail -1)" | tee -a $L
echo "stress16: $(tools/microbench/stress16.sh 233217 600 16 2>&1 | head -1)" | tee -a $L
