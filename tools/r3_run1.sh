#!/bin/bash
# round 3, visit 1: lane-utilisation counters on both scenes + first A/B of the cheap secondary-scene fixes
set -u
OUT=gpurun_out/r3_run1
mkdir -p $OUT
export TMPDIR=/tmp
echo "== subset tests ==" | tee $OUT/summary.log
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "color or colour or slice or plain_path or spherical or reduce or tuple" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log | tee -a $OUT/summary.log
echo "== lane stats ==" | tee -a $OUT/summary.log
timeout 600 python tools/lane_stats.py --scene both 2>&1 | grep '^{' | tee $OUT/lane_stats.jsonl | cut -c1-1500 | tee -a $OUT/summary.log
echo "== bench ==" | tee -a $OUT/summary.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log > $OUT/bench.json
python - <<'PY' | tee -a $OUT/summary.log
import json
d=json.load(open('gpurun_out/r3_run1/bench.json'))
print('headline', d['value'], d['ms_per_step'], d['stage_ms'])
s=d['config']['secondary']
print('secondary', s['value'], s['ms_per_step'], s['stage_ms'], s['depth_slices'])
PY
echo "== done ==" | tee -a $OUT/summary.log
