#!/bin/bash
# round-6 GPU visits: bash tools/visit_r6.sh <tag> <step...>; steps as tools/gpu_visit.sh plus:
#   r6tests           tests/test_gpu_round6.py
#   sel_ab            headline + configs with GSD_DEPTH_SELECT=1 (default) against 0, interleaved
TAG=${1:-r6}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT
S=$OUT/summary_r6.log; : > $S
line() { python - "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith('{'):
        d = json.loads(l); c = d['config']
        print(sys.argv[1], 'ms', d['ms_per_step'], 'stages', d.get('stage_ms'), 'slices', c.get('depth_slices'), 'hints', c.get('frame_hints'))
PY
}
rest=()
for step in "$@"; do
  case $step in
    r6tests)
      timeout 1500 python -m pytest tests/test_gpu_round6.py -m gpu -q -s -x -p no:cacheprovider --timeout 600 --tb=short > $OUT/pytest_r6.log 2>&1
      grep -E "^nearest|^FAILED|^ERROR|passed|failed|Error|assert" $OUT/pytest_r6.log | tail -40 | tee -a $S ;;
    sel_ab)
      for v in 1 2; do for sel in 1 0; do
        GSD_DEPTH_SELECT=$sel timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-view-sweep > $OUT/h_s${sel}_$v.log 2>&1; line "headline sel=$sel" $OUT/h_s${sel}_$v.log | tee -a $S
        GSD_DEPTH_SELECT=$sel timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-view-sweep --gaussians 1000000 --subposes 1 --rs-bands 10 > $OUT/c3_s${sel}_$v.log 2>&1; line "config3 sel=$sel" $OUT/c3_s${sel}_$v.log | tee -a $S
      done; done
      for sel in 1 0; do
        GSD_DEPTH_SELECT=$sel timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-view-sweep --gaussians 2000000 --subposes 5 --rs-bands 2 > $OUT/c4_s${sel}.log 2>&1; line "config4 sel=$sel" $OUT/c4_s${sel}.log | tee -a $S
        GSD_DEPTH_SELECT=$sel timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-view-sweep --gaussians 5000000 --width 3840 --height 2160 --subposes 10 > $OUT/c5_s${sel}.log 2>&1; line "config5 sel=$sel" $OUT/c5_s${sel}.log | tee -a $S
      done ;;
    *) rest+=("$step") ;;
  esac
done
if [ ${#rest[@]} -gt 0 ]; then bash tools/gpu_visit.sh $TAG "${rest[@]}"; cat $OUT/summary.log >> $S; fi
