#!/bin/bash
# full GPU suite + smoke + bench (+ optional A/B of the Python orchestration)
set -u
OUT=gpurun_out/${1:-r3_full}
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for v in 1 0; do
  GSD_NATIVE_FRAME=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_native$v.log 2>&1
  grep '^{' $OUT/bench_native$v.log > $OUT/bench_native$v.json
  python - $OUT/bench_native$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('native=%s headline' % sys.argv[2], d['value'], d['ms_per_step'], d['stage_ms'])
s = d['config']['secondary']
print('native=%s secondary' % sys.argv[2], s['value'], s['ms_per_step'], s['stage_ms'])
PY
done
