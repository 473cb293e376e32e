#!/bin/bash
# verification of the committed tree: full GPU suite, smoke, default bench.py (with the cpu baseline)
set -u
OUT=gpurun_out/${1:-r3_final}
mkdir -p $OUT
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('valu'), d['roofline'].get('traffic'))
s = d['config']['secondary']
print('secondary', s['value'], s['ms_per_step'], s['depth_slices'], s['train_step'])
print('cpu', d.get('cpu_baseline'))
PY
