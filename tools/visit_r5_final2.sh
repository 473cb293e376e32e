# last visit of round 5 (second edition, after the packed counts / self-zeroing bin edges / lazy records): the full suite
# (six xdist workers share the GPU — the oracle comparisons are CPU-bound; the one wall-clock test runs alone afterwards),
# smoke, kernel table + timeline + PMC passes of the committed tree, then the driver's command with the fresh counters
set -u
OUT=gpurun_out/r5_final2; mkdir -p $OUT
ALT="tests/test_gpu_parity.py::test_alternating_scenes_of_one_shape_keep_their_frame_time"
# (as first run, WITHOUT the thread caps below, the six workers' float64 oracle comparisons oversubscribed the host's
#  cores sixfold: 132 of 237 tests in 12 minutes, all green, and the round's GPU budget ended there — serial it is 9.5)
OMP_NUM_THREADS=2 MKL_NUM_THREADS=2 timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider -n 6 --timeout 900 --tb=short --deselect $ALT > $OUT/pytest_gpu.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -20
timeout 300 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 300 --tb=short -k "alternating_scenes" > $OUT/pytest_alt.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_alt.log | tail -3
bash tools/gpu_visit.sh r5_final2b smoke prof pmc 2>&1 | grep -v "^void\|^gs::\|^\"" | tail -30
cp gpurun_out/r5_final2b/traffic.json profiles/traffic.json
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.log 2> $OUT/driver_cmd.err; echo "driver command rc=$?"
grep '^{' $OUT/driver_cmd.log > $OUT/final_bench.json
python - $OUT/final_bench.json <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read()); s=d['config']['secondary']; r=d['roofline']
print('headline', d['ms_per_step'], d['value'], 'stall', d['host_stall_ms'], d['timing_attempts_ms'], d['host_stall_check'], d['stage_ms'])
print('roofline frac', r['frac'], 'achieved', r['achieved'], 'traffic', r['traffic'], 'kernel ms', r['kernel_ms_per_step'], 'valu', r['valu'] and {k: r['valu'][k] for k in ('issue_frac_at_2_cycles','issue_frac_at_measured_mix','waves_per_simd')})
print('secondary', s['ms_per_step'], 'stall', s['host_stall_ms'], s['timing_attempts_ms'], s['depth_slices'], s['frame_hints'], s['train_step'], 'roofline', s['roofline']['frac'])
print('cpu_baseline', d.get('cpu_baseline'))
PY
