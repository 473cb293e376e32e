# last visit of round 5: full suite, smoke, kernel table + PMC passes of the committed tree, then the driver's command
# with the fresh counters in place, the exchange over RCCL at world 1, and the three motion models
set -u
OUT=gpurun_out/r5_final; mkdir -p $OUT
bash tools/gpu_visit.sh r5_final suite smoke prof pmc
cp $OUT/traffic.json profiles/traffic.json
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.log 2> $OUT/driver_cmd.err; echo "driver command rc=$?" | tee -a $OUT/summary.log
grep '^{' $OUT/driver_cmd.log > $OUT/final_bench.json
python - $OUT/final_bench.json <<'PY' | tee -a $OUT/summary.log
import json, sys
d=json.loads(open(sys.argv[1]).read()); s=d['config']['secondary']; r=d['roofline']
print('headline', d['ms_per_step'], d['value'], 'stall', d['host_stall_ms'], d['timing_attempts_ms'], d['host_stall_check'])
print('roofline frac', r['frac'], 'achieved', r['achieved'], 'traffic', r['traffic'], 'kernel ms', r['kernel_ms_per_step'], 'valu', r['valu'] and {k: r['valu'][k] for k in ('issue_frac_at_2_cycles','issue_frac_at_measured_mix','waves_per_simd')})
print('secondary', s['ms_per_step'], 'stall', s['host_stall_ms'], s['timing_attempts_ms'], s['depth_slices'], s['frame_hints'], s['train_step'], 'roofline', s['roofline']['frac'])
print('cpu_baseline', d.get('cpu_baseline'))
PY
bash tools/gpu_visit.sh r5_final_extra prof_exchange motions abflag:--autograd
