#!/bin/bash
# A/B of the compositor variants inside one GPU-box visit: 0 = scalar-cache (s_load) kernels, 2 = round-1
# v_readlane kernels.  Interleaved rounds, bench.py stage table.
OUT=gpurun_out; mkdir -p $OUT
for round in 1 2; do
  for v in "0 0" "0 2" "2 2"; do
    set -- $v
    GSD_RASTER_FWD_VARIANT=$1 GSD_RASTER_BWD_VARIANT=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['stage_ms']; print('fwd/bwd variant $v round $round:', d['value'], 'MPix/s', d['ms_per_step'], 'ms  raster_fwd', s.get('raster_fwd'), 'raster_bwd', s.get('raster_bwd'), 'bin_edges', s.get('bin_edges'), 'grad_reduce', s.get('grad_reduce'))" | tee -a $OUT/ab_fwd.log
  done
done
