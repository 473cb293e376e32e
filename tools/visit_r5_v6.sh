set -u
OUT=gpurun_out/r5_v6; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 --tb=short --durations=25 > $OUT/pytest_gpu.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|^E  |pose optimizer, gradient" $OUT/pytest_gpu.log | tail -30
grep -A26 "slowest" $OUT/pytest_gpu.log | head -30
python tools/fragile_table.py $OUT/pytest_gpu.log > $OUT/fragile_table.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.log 2> $OUT/driver_cmd.err; echo bench rc=$?
python - $OUT/driver_cmd.log <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); s=d['config']['secondary']
        print('headline', d['ms_per_step'], d['value'], 'stall', d['host_stall_ms'], d['timing_attempts_ms'], d['host_stall_check'], 'stages', d['stage_ms'])
        print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','kernel_ms_per_step','traffic')}, 'valu', d['roofline']['valu'])
        print('secondary', s['ms_per_step'], 'stall', s['host_stall_ms'], s['timing_attempts_ms'], s['depth_slices'], s['frame_hints'], s['train_step'], 'stages', s['stage_ms'])
PY
