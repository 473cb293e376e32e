#!/bin/bash
# fitted-model-like scene: sensitivity to the depth-slice budget (few tiles saturate there, so slicing buys little)
set -u
OUT=gpurun_out/r3_run12
mkdir -p $OUT
for b in 512 1024 2048 0; do
  GSD_SLICE_BASE=$b timeout 300 python bench.py --scene trained --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_b$b.log 2>&1
  grep '^{' $OUT/bench_b$b.log > $OUT/bench_b$b.json
  python - $OUT/bench_b$b.json $b <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('slice_base=%s' % sys.argv[2], d['value'], d['ms_per_step'], d['config']['depth_slices'], d['stage_ms'])
PY
done
