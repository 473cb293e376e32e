# lazy records (GSD_LAZY_RECORDS=1, default: scenes whose frames stop within the default budget) against the eager
# projection (0): the bit-equality test + the tests downstream of the projection, then headline / config 3 / config 4
# interleaved, and the default bench line with the fitted-model-like scene (its budget grows: it must stay eager)
set -u
OUT=gpurun_out/r5_lazy; mkdir -p $OUT
timeout 800 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "lazy_records or frame or orchestration or knob or golden or full_size_headline_vs or band_aware or speculative or routes_agree or render_step or raw_param or alternating" 2>&1 | tail -8
for v in 1 2; do
 for cg in 1 0; do
  GSD_LAZY_RECORDS=$cg timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/h_p${cg}_$v.log 2>&1
  GSD_LAZY_RECORDS=$cg timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --gaussians 1000000 --subposes 1 --rs-bands 10 > $OUT/c3_p${cg}_$v.log 2>&1
  GSD_LAZY_RECORDS=$cg timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --gaussians 2000000 --subposes 5 --rs-bands 2 > $OUT/c4_p${cg}_$v.log 2>&1
  python - $OUT/h_p${cg}_$v.log $OUT/c3_p${cg}_$v.log $OUT/c4_p${cg}_$v.log "lazy=$cg round $v" <<'PY'
import json, sys
for f in sys.argv[1:4]:
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); print(sys.argv[4], 'N=%d R=%d' % (d['config']['gaussians'], d['config']['rs_bands']), 'ms', d['ms_per_step'], 'project_fwd', d['stage_ms']['project_fwd'], 'slice_count', d['stage_ms']['slice_count'], 'project_bwd', d['stage_ms']['project_bwd'])
PY
 done
done
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/full.log 2>&1
python - $OUT/full.log <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); s=d['config']['secondary']
        print('default line: headline', d['ms_per_step'], 'secondary', s['ms_per_step'], 'stall', s['host_stall_ms'], 'project_fwd', s['stage_ms']['project_fwd'], 'slice_count', s['stage_ms']['slice_count'], s['frame_hints'], s['train_step'])
PY
