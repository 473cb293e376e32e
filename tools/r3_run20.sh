#!/bin/bash
# fuzz of the final tree: default path (random routes, C++ frame with merging / polling, both sort forms) vs plain path,
# and a batch against the float64 oracle
set -u
OUT=gpurun_out/r3_run20
mkdir -p $OUT
for seed in 31 32 33; do
  timeout 600 python tests/fuzz_paths.py 1200 $seed > $OUT/fuzz_$seed.log 2>&1
  echo "seed $seed: $(tail -1 $OUT/fuzz_$seed.log)"; grep FAIL $OUT/fuzz_$seed.log | head -5
done
timeout 900 python tests/fuzz_paths.py 200 41 oracle > $OUT/fuzz_oracle.log 2>&1
echo "oracle seed 41: $(tail -1 $OUT/fuzz_oracle.log)"; grep FAIL $OUT/fuzz_oracle.log | head -5
