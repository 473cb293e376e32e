set -u
OUT=gpurun_out/r5_v4; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 --tb=short --durations=15 -k "upstream_gradient or fork_style or baseline_configs or golden or rolling or config3 or config4 or full_size or equals_plain or frame" > $OUT/pytest_sel.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|what the conventions|opaque splats|fork keywords|^E  " $OUT/pytest_sel.log | tail -40
grep -A16 "slowest" $OUT/pytest_sel.log | head -20
for f in 7 6 3 0; do
  GSD_UPSTREAM_GRADS=$f timeout 300 python -m pytest tests/test_data_and_training.py -m gpu -q -s -p no:cacheprovider -k "pose_optimizer" > $OUT/pose_$f.log 2>&1
  echo "== pose optimizer, GSD_UPSTREAM_GRADS=$f"; grep -E "^frame|passed|failed" $OUT/pose_$f.log
done
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
python tools/build_variant.py raster_bwd.hip /tmp/libgsd_exec.so -DGS_BWD_EXEC=1 > $OUT/abbuild.log 2>&1
for v in 1 2; do
  for lib in base exec; do
    if [ $lib = exec ]; then export GSD_LIB_PATH=/tmp/libgsd_exec.so; else unset GSD_LIB_PATH; fi
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/ab_${lib}$v.log 2>&1
    python - $OUT/ab_${lib}$v.log "bwd $lib $v" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); s=d['config']['secondary']; print(sys.argv[2], 'headline ms', d['ms_per_step'], 'bwd', d['stage_ms']['raster_bwd'], '| secondary ms', s['ms_per_step'], 'bwd', s['stage_ms']['raster_bwd'], 'train', s['train_step'])
PY
  done
done
unset GSD_LIB_PATH
