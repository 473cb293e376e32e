#!/bin/bash
# round-6 GPU visits: bash tools/visit_r6.sh <tag> <step...>; steps as tools/gpu_visit.sh plus:
#   r6tests           tests/test_gpu_round6.py
#   sel_ab            headline + configs with GSD_DEPTH_SELECT=1 (default) against 0, interleaved
#   c5x3              config 5's share three times on one box (stall hunt)
#   merge_ab          the view sweep with everything behind an open slice issued as ONE slice (patch build; MERGE_PATCH=<name> picks
#                     another patch of tools/patches/) against the doubling spans
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
    splat_ab)
      # the splat-parallel backward (GSD_BWD_SPLAT=1) against the tile-per-wave kernel, headline and fitted-model-like scene
      for sc in survey trained; do for v in 0 1; do
        GSD_BWD_SPLAT=$v timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-view-sweep --scene $sc > $OUT/splat${v}_$sc.log 2>&1; line "$sc splat=$v" $OUT/splat${v}_$sc.log | tee -a $S
      done; done
      R=${GRAFT_REPO_ROOT:-$(pwd)}
      (cd /tmp && GSD_BWD_SPLAT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_splat -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-view-sweep --scene trained) > $OUT/prof_splat.log 2>&1
      for f in $(find $OUT/prof_splat -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/kernel_stats_splat_trained.csv; head -8 $f | cut -c1-220 | tee -a $S; done
      rm -rf $OUT/prof_splat/*/*kernel_trace* 2>/dev/null ;;
    nored_ab)
      # upper bound of ANY rewrite of the backward's per-entry reduction: a build whose bwd_entry computes the nine sums
      # and reduces / stores nothing (tools/patches/bwd_no_reduction.json; gradients are wrong, times are the point)
      python tools/ab_patch.py tools/patches/bwd_no_reduction.json /tmp/libgsd_nored.so > $OUT/nored_build.log 2>&1 || { tail -3 $OUT/nored_build.log | tee -a $S; }
      for sc in survey trained; do for v in 1 2; do
        timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-view-sweep --scene $sc > $OUT/nored_base${v}_$sc.log 2>&1; line "$sc base$v" $OUT/nored_base${v}_$sc.log | tee -a $S
        GSD_LIB_PATH=/tmp/libgsd_nored.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-view-sweep --scene $sc > $OUT/nored_alt${v}_$sc.log 2>&1; line "$sc no-reduction$v" $OUT/nored_alt${v}_$sc.log | tee -a $S
      done; done ;;
    nostore_ab)
      # how much of the backward's reduction cost is the tuple STORES (36 scattered bytes per touched entry): a build that
      # reduces in LDS as the product does and stores nothing (tools/patches/bwd_no_tuple_stores.json)
      python tools/ab_patch.py tools/patches/bwd_no_tuple_stores.json /tmp/libgsd_nostore.so > $OUT/nostore_build.log 2>&1 || { tail -3 $OUT/nostore_build.log | tee -a $S; }
      for sc in survey trained; do for v in 1 2; do
        timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-view-sweep --scene $sc > $OUT/nostore_base${v}_$sc.log 2>&1; line "$sc base$v" $OUT/nostore_base${v}_$sc.log | tee -a $S
        GSD_LIB_PATH=/tmp/libgsd_nostore.so timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-view-sweep --scene $sc > $OUT/nostore_alt${v}_$sc.log 2>&1; line "$sc no-tuple-stores$v" $OUT/nostore_alt${v}_$sc.log | tee -a $S
      done; done ;;
    budget_ab)
      # first-slice budget on the large configurations: does a smaller budget + a second slice pay where the tile sort is big?
      for base in 512 384 256; do
        GSD_SLICE_BASE=$base timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-view-sweep --gaussians 5000000 --width 3840 --height 2160 --subposes 10 > $OUT/c5_b$base.log 2>&1; line "config5 base=$base" $OUT/c5_b$base.log | tee -a $S
        GSD_SLICE_BASE=$base timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-view-sweep --gaussians 2000000 --subposes 5 --rs-bands 2 > $OUT/c4_b$base.log 2>&1; line "config4 base=$base" $OUT/c4_b$base.log | tee -a $S
      done ;;
    merge_ab)
      # how much of a multi-slice frame is slice boundaries: a build that issues everything behind a slice that left tiles
      # open as ONE slice (tools/patches/merge_rest_after_first.json) against the doubling spans, on the view sweep
      python tools/ab_patch.py tools/patches/${MERGE_PATCH:-merge_rest_after_first}.json /tmp/libgsd_merge.so > $OUT/merge_build.log 2>&1 || { tail -3 $OUT/merge_build.log | tee -a $S; }
      for v in 1 2; do for lib in base merge; do
        env $( [ $lib = merge ] && echo GSD_LIB_PATH=/tmp/libgsd_merge.so ) timeout 400 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > $OUT/merge_${lib}$v.log 2>&1
        python - $OUT/merge_${lib}$v.log $lib$v <<'PY' | tee -a $S
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l); v = d['config']['view_sweep']; s = d['config']['secondary']
        print(sys.argv[2], 'headline', d['ms_per_step'], 'secondary', s['ms_per_step'], s['depth_slices'], 'sweep per-view', v['ms_per_view'], 'fixed', v['fixed_view_ms'],
              'slices', v['slices_per_frame'], 'one-memory', v['one_memory_for_all_cameras']['ms_per_view'])
PY
      done; done ;;
    c5x3)
      # config 5's share on one GPU, three times on one box (visit r6_final2 saw ONE first attempt at 48.7 ms against 12.2 in
      # its retry and in every other visit): ms, attempts, and the per-step host issue times of a stalled attempt
      for v in 1 2 3; do
        timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-view-sweep --gaussians 5000000 --width 3840 --height 2160 --subposes 10 > $OUT/c5_run$v.log 2>&1
        python - $OUT/c5_run$v.log <<'PY' | tee -a $S
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print('config5 run', sys.argv[1][-5], 'ms', d['ms_per_step'], 'attempts', d['timing_attempts_ms'], 'stall', d['host_stall_ms'],
              'issue', d.get('host_issue_ms_per_step'), 'stages', d['stage_ms'])
PY
      done ;;
    *) rest+=("$step") ;;
  esac
done
if [ ${#rest[@]} -gt 0 ]; then bash tools/gpu_visit.sh $TAG "${rest[@]}"; cat $OUT/summary.log >> $S; fi
