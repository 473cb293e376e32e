#!/bin/bash
# A/B of run-time knobs (environment variables of ops.py), interleaved, same build.
# usage: bash tools/ab_env.sh VAR v1 v2 ...
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
VAR=$1; shift
for rep in 1 2; do
for v in "$@"; do
  env $VAR=$v timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$VAR=$v', d['value'], d['ms_per_step'], d['config'].get('depth_slices'), {k:d['stage_ms'].get(k) for k in ('raster_fwd','raster_bwd','depth_sort','count_scan','slice_plan','tile_sort','emit','slice_count')})" | tee -a gpurun_out/ab_env.log
done; done
