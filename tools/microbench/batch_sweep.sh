for cfg in "fhd 1 4" "fhd 4 2" "fhd 8 2" "fhd 16 1" "fhd 16 2" "fhd 16 3" "4k 1 4" "4k 4 2" "4k 8 1" "4k 8 2" "8k 1 3" "8k 2 2" "8k 4 1" "8k 4 2"; do
  timeout 120 python tools/microbench/batch_throughput.py $cfg 1.5
done
EFX_NO_BATCH=1 timeout 120 python tools/microbench/batch_throughput.py fhd 16 2 1.5
