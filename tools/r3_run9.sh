#!/bin/bash
set -u
OUT=gpurun_out/r3_run9
mkdir -p $OUT
# correctness of the lock-step forward: force it everywhere (variant 3) and run the path-equivalence / oracle tests
GSD_RASTER_FWD_VARIANT=3 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "plain_path or native_frame or golden or baseline_configs or fused or depth or model" > $OUT/pytest_quad.log 2>&1
tail -5 $OUT/pytest_quad.log
for v in 0 4; do
  GSD_RASTER_FWD_VARIANT=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_v$v.log 2>&1
  grep '^{' $OUT/bench_v$v.log > $OUT/bench_v$v.json
  python - $OUT/bench_v$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('fwd_variant=%s headline' % sys.argv[2], d['value'], d['ms_per_step'], d['stage_ms']['raster_fwd'])
s = d['config']['secondary']
print('fwd_variant=%s secondary' % sys.argv[2], s['value'], s['ms_per_step'], s['stage_ms']['raster_fwd'])
PY
done
