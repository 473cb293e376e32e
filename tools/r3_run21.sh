#!/bin/bash
set -u
OUT=gpurun_out/r3_run21
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "native_frame" 2>&1 | tail -4
for seed in 31 32; do
  timeout 900 python tests/fuzz_paths.py 1200 $seed > $OUT/fuzz_$seed.log 2>&1
  echo "seed $seed: $(tail -1 $OUT/fuzz_$seed.log)"; grep FAIL $OUT/fuzz_$seed.log | head -5
done
