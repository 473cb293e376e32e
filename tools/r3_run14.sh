#!/bin/bash
# polled read-backs (gs_frame_desc.poll_readback): the frame tests, then both scenes with / without, twice interleaved
set -u
OUT=gpurun_out/r3_run14
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "native_frame or exact_rolling or depth_sliced" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for rep in 1 2; do
for m in 0 1; do
  GSD_FRAME_POLL=$m timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_p${m}_$rep.log 2>&1
  grep '^{' $OUT/bench_p${m}_$rep.log > $OUT/bench_p${m}_$rep.json
  python - $OUT/bench_p${m}_$rep.json $m <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
s = d['config']['secondary']
print('poll=%s headline' % sys.argv[2], d['value'], d['ms_per_step'], ' secondary', s['value'], s['ms_per_step'], s['depth_slices'])
PY
done
done
