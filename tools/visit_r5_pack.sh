# packed tile counts in the depth pre-sort's payload (GSD_COMPACT_PACK=1, default) against the random gather (0):
# sort / frame / golden tests under the default, then headline, config 3 and config 4 interleaved
set -u
OUT=gpurun_out/r5_pack; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "sort or compacting or binning or frame or orchestration or knob or golden or full_size_headline_vs or band_aware or speculative" 2>&1 | tail -4
for v in 1 2; do
 for cg in 1 0; do
  GSD_COMPACT_PACK=$cg timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/h_p${cg}_$v.log 2>&1
  GSD_COMPACT_PACK=$cg timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --gaussians 1000000 --subposes 1 --rs-bands 10 > $OUT/c3_p${cg}_$v.log 2>&1
  GSD_COMPACT_PACK=$cg timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --gaussians 2000000 --subposes 5 --rs-bands 2 > $OUT/c4_p${cg}_$v.log 2>&1
  python - $OUT/h_p${cg}_$v.log $OUT/c3_p${cg}_$v.log $OUT/c4_p${cg}_$v.log "pack=$cg round $v" <<'PY'
import json, sys
for f in sys.argv[1:4]:
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); print(sys.argv[4], 'N=%d R=%d' % (d['config']['gaussians'], d['config']['rs_bands']), 'ms', d['ms_per_step'], 'depth_sort', d['stage_ms']['depth_sort'], 'count_scan', d['stage_ms']['count_scan'])
PY
 done
done
