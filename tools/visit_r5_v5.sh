set -u
OUT=gpurun_out/r5_v5; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 --tb=short --durations=8 -k "upstream_gradient or fork_style or config2 or golden or equals_plain or orchestration or band_aware or pose_optimizer or config3 or config4" > $OUT/pytest_sel.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|what the conventions|fork keywords|band-aware|^frame |^E  " $OUT/pytest_sel.log | tail -40
grep -A9 "slowest" $OUT/pytest_sel.log | head -12
for ba in 1 0; do
 for cfg in "c3 --gaussians 1000000 --subposes 1 --rs-bands 10" "c4 --gaussians 2000000 --subposes 5 --rs-bands 2"; do
  name=${cfg%% *}; fl=${cfg#* }
  GSD_BAND_AWARE=$ba timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $fl > $OUT/bench_${name}_ba$ba.log 2>&1
  python - $OUT/bench_${name}_ba$ba.log "$name band_aware=$ba" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], 'ms', d['ms_per_step'], 'stall', d.get('host_stall_ms'), 'stages', d['stage_ms'], 'I', d['config']['tile_intersections_per_step'], d['config']['depth_slices'])
PY
  grep -E "Error|error" $OUT/bench_${name}_ba$ba.log | tail -2
 done
done
