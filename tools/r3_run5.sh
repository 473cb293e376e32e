#!/bin/bash
set -u
OUT=gpurun_out/r3_run5
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "slice or plain_path or binning or emit or count or fuzz or multi" > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log > $OUT/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_run5/bench.json'))
print('headline', d['value'], d['ms_per_step'], d['stage_ms'])
s=d['config']['secondary']
print('secondary', s['value'], s['ms_per_step'], s['stage_ms'], s['depth_slices'])
PY
