#!/bin/bash
# single-pass (decoupled look-back) radix passes: the sort / binning / frame tests, then A/B of both scenes
set -u
OUT=gpurun_out/r3_run15
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "sort or radix or binning or scan or native_frame or runtime_knob or depth_sliced or workspace" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for rep in 1 2; do
for m in 0 1; do
  GSD_SORT_SINGLE_PASS=$m timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_s${m}_$rep.log 2>&1
  grep '^{' $OUT/bench_s${m}_$rep.log > $OUT/bench_s${m}_$rep.json
  python - $OUT/bench_s${m}_$rep.json $m <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
s = d['config']['secondary']
print('single_pass=%s headline' % sys.argv[2], d['value'], d['ms_per_step'], 'depth_sort', d['stage_ms']['depth_sort'], 'tile_sort', d['stage_ms']['tile_sort'],
      ' secondary', s['value'], s['ms_per_step'], 'depth_sort', s['stage_ms']['depth_sort'], 'tile_sort', s['stage_ms']['tile_sort'])
PY
done
done
