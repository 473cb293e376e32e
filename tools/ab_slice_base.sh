#!/bin/bash
mkdir -p gpurun_out/r4_slice
for v in 1 2; do
for sb in ${SB_LIST:-512 1024 2048}; do
  GSD_SLICE_ADAPT=0 GSD_SLICE_BASE=$sb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --scene trained > gpurun_out/r4_slice/sb${sb}_$v.log 2>&1
  python - $sb gpurun_out/r4_slice/sb${sb}_$v.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith('{'):
        d = json.loads(l); s = d['stage_ms']
        print('slice_base', sys.argv[1], 'ms', d['ms_per_step'], 'slices', d['config']['depth_slices'], 'fwd', s.get('raster_fwd'), 'bwd', s.get('raster_bwd'), 'count', s.get('slice_count'), 'emit', s.get('emit'), 'tsort', s.get('tile_sort'), 'reduce', s.get('grad_reduce'))
PY
done; done
