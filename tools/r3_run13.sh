#!/bin/bash
# slice merging (gs_frame_desc.merge_open_fraction): tests that touch the slice loop, then both scenes with / without it
set -u
OUT=gpurun_out/r3_run13
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "native_frame or depth_sliced or depth_channel or multi_slice or runtime_knob or speculat" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for m in 0 0.75; do
  GSD_SLICE_MERGE=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_m$m.log 2>&1
  grep '^{' $OUT/bench_m$m.log > $OUT/bench_m$m.json
  python - $OUT/bench_m$m.json $m <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
s = d['config']['secondary']
print('merge=%s headline' % sys.argv[2], d['value'], d['ms_per_step'], d['config']['depth_slices'])
print('merge=%s secondary' % sys.argv[2], s['value'], s['ms_per_step'], s['depth_slices'], s['stage_ms'])
PY
done
