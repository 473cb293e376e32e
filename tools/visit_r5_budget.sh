set -u
OUT=gpurun_out/r5_budget; mkdir -p $OUT
for sb in 192 256 384 512; do
  GSD_SLICE_ADAPT=0 GSD_SLICE_BASE=$sb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/sb$sb.log 2>&1
  python - $OUT/sb$sb.log "budget $sb" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], 'ms', d['ms_per_step'], 'slices', d['config']['depth_slices'], 'stages', d['stage_ms'])
PY
done
