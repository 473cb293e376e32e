#!/bin/bash
set -u
OUT=gpurun_out/r3_run22
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "native_frame or runtime_knob" 2>&1 | tail -3
for rep in 1 2 3; do
for m in 0 1; do
  GSD_PREALLOC_BWD=$m timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_${m}_$rep.log 2>&1
  grep '^{' $OUT/bench_${m}_$rep.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('prealloc=$m', d['value'], d['ms_per_step'])"
done
done
