#!/bin/bash
# One GPU-box visit, parametrised (replaces the round-3 one-off recipes).  Usage, from the repo root on the GPU box:
#   bash tools/gpu_visit.sh <tag> [steps...]      steps: suite | tests:<pytest -k expr> | smoke | bench | benchq | ab:<ENV=V,...>
#                                                        | prof | pmc | pmc_trained | prof_trained | prof_exchange | motions | fuzz | configs
# Everything lands in gpurun_out/<tag>/; a summary is printed at the end.
set -u
TAG=${1:-visit}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
S=$OUT/summary.log
: > $S
ABB_ARGS=${ABB_ARGS---no-secondary --no-view-sweep}      # extra bench.py arguments of the abbuild step (default: headline only)
benchline() { python - "$1" "$2" <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
for l in open(path):
    if l.startswith('{'):
        d = json.loads(l)
        sec = d['config'].get('secondary') or {}
        print(tag, 'ms', d['ms_per_step'], 'MPix/s', d['value'], 'stages', d.get('stage_ms'), 'slices', d['config'].get('depth_slices'))
        if sec:
            print(tag, 'secondary ms', sec.get('ms_per_step'), 'stages', sec.get('stage_ms'), 'slices', sec.get('depth_slices'))
PY
}
for step in "$@"; do
  echo "== $step ==" | tee -a $S
  case $step in
    suite)
      timeout 1700 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 --tb=short > $OUT/pytest_gpu.log 2>&1
      grep -E "^\[kernel-vs|^\[fragile|^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -120 | tee -a $S ;;
    tests:*)
      timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 --tb=short -k "${step#tests:}" > $OUT/pytest_sel.log 2>&1
      grep -E "^\[kernel-vs|^\[fragile|^FAILED|^ERROR|passed|failed" $OUT/pytest_sel.log | tail -80 | tee -a $S ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log | tee -a $S ;;
    bench)
      timeout 600 python bench.py > $OUT/bench.log 2>&1; grep '^{' $OUT/bench.log > $OUT/bench.json; benchline bench $OUT/bench.json | tee -a $S ;;
    benchq)
      for v in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/benchq$v.log 2>&1; benchline benchq$v $OUT/benchq$v.log | tee -a $S; done ;;
    abbuild:*)
      # A/B of a compile-time variant: abbuild:<file.hip>:<-DFLAG=V>[:<-DFLAG2>]  (tools/build_variant.py)
      spec="${step#abbuild:}"; f="${spec%%:*}"; defs=$(echo "${spec#*:}" | tr ':' ' ')
      alt=/tmp/libgsd_$(echo "$spec" | tr -c 'A-Za-z0-9' '_').so
      python tools/build_variant.py $f $alt $defs > $OUT/abbuild.log 2>&1 || { tail -5 $OUT/abbuild.log | tee -a $S; continue; }
      for v in 1 2; do
        timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $ABB_ARGS > $OUT/abb_base$v.log 2>&1; benchline base$v $OUT/abb_base$v.log | tee -a $S
        GSD_LIB_PATH=$alt timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $ABB_ARGS > $OUT/abb_alt$v.log 2>&1; benchline "alt$v($defs)" $OUT/abb_alt$v.log | tee -a $S
      done ;;
    motions)
      # the three motion models on the headline scene and on the fitted-model-like one
      for m in se3 pixel_velocity pixel_velocity_shared; do
        timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --motion $m > $OUT/motion_$m.log 2>&1; benchline $m $OUT/motion_$m.log | tee -a $S
        grep -E "Error|error" $OUT/motion_$m.log | tail -3 | tee -a $S
      done ;;
    abflag:*)
      # A/B of a bench.py command-line flag (e.g. abflag:--autograd), interleaved
      fl="${step#abflag:}"
      for v in 1 2; do
        timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-view-sweep > $OUT/abf_base$v.log 2>&1; benchline base$v $OUT/abf_base$v.log | tee -a $S
        timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-view-sweep $fl > $OUT/abf_alt$v.log 2>&1; benchline "alt$v($fl)" $OUT/abf_alt$v.log | tee -a $S
      done ;;
    ab:*)
      envs=$(echo "${step#ab:}" | tr ',' ' ')
      for v in 1 2; do
        timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-view-sweep > $OUT/ab_base$v.log 2>&1; benchline base$v $OUT/ab_base$v.log | tee -a $S
        env $envs timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-view-sweep > $OUT/ab_alt$v.log 2>&1; benchline "alt$v($envs)" $OUT/ab_alt$v.log | tee -a $S
      done ;;
    prof|prof_trained)
      extra=""; [ $step = prof_trained ] && extra="--scene trained"
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/$step -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-view-sweep $extra) > $OUT/$step.log 2>&1
      for f in $(find $OUT/$step -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/kernel_stats_$step.csv; head -24 $f | cut -c1-200 | tee -a $S; done
      [ $step = prof ] && python tools/trace_step.py $OUT/prof > $OUT/timeline.txt 2>/dev/null && tail -1 $OUT/timeline.txt | tee -a $S
      rm -rf $OUT/$step/*/*kernel_trace* 2>/dev/null ;;
    configs)
      # bench lines of the other BASELINE.json configurations on one GPU (their parity is tested by the suite; these are
      # measurements only): config 3 = 1M, 1080p, 10 rolling-shutter row bands; config 4's per-GPU share = 2M, S=5 x R=2;
      # config 5's per-GPU share = 5M, 3840x2160, S=10
      for cfg in "c3 --gaussians 1000000 --subposes 1 --rs-bands 10" "c4 --gaussians 2000000 --subposes 5 --rs-bands 2" "c5 --gaussians 5000000 --width 3840 --height 2160 --subposes 10"; do
        name=${cfg%% *}; fl=${cfg#* }
        timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-view-sweep $fl > $OUT/bench_$name.log 2>&1
        benchline "$name($fl)" $OUT/bench_$name.log | tee -a $S
        grep -E "Error|error" $OUT/bench_$name.log | tail -2 | tee -a $S
      done ;;
    fuzz)
      # randomised equivalence / oracle checks (tests/fuzz_paths.py): default path vs the plain one over random sizes and
      # switches; with `oracle` tiny scenes are also held against the float64 oracle; `pixvel`: the pixel-velocity model
      # (a third of those trials with exact rolling shutter).  Each leg stops after 140 s (a killed leg prints no total).
      for leg in "400 41 oracle" "400 42" "300 43 oracle pixvel" "200 44 pixvel"; do
        tag=$(echo $leg | tr ' ' '_')
        timeout 140 python tests/fuzz_paths.py $leg > $OUT/fuzz_$tag.log 2>&1
        echo "fuzz [$leg]: $(grep -c ' ok$' $OUT/fuzz_$tag.log) ok, $(grep -c 'FAIL$' $OUT/fuzz_$tag.log) FAIL; $(tail -1 $OUT/fuzz_$tag.log | cut -c1-120)" | tee -a $S
        grep 'FAIL$' $OUT/fuzz_$tag.log | head -3 | cut -c1-300 | tee -a $S
      done ;;
    prof_exchange)
      # the DP gradient exchange over RCCL at world size 1 (bench.py --force-exchange): bench line + kernel table with
      # the ncclDevKernel rows
      timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-view-sweep --force-exchange > $OUT/bench_exchange.log 2>&1
      python - $OUT/bench_exchange.log <<'PY' | tee -a $S
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l); c = d['config']
        print('force-exchange: ms', d['ms_per_step'], 'exchange_ms', d.get('exchange_ms'), 'mode', c.get('gradient_exchange'), 'rccl', c.get('rccl_version'), 'per_rank', c.get('per_rank'))
PY
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_exchange -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-view-sweep --force-exchange) > $OUT/prof_exchange.log 2>&1
      for f in $(find $OUT/prof_exchange -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/kernel_stats_prof_exchange.csv; grep -i "nccl\|dp_" $f | cut -c1-160 | tee -a $S; done
      rm -rf $OUT/prof_exchange/*/*kernel_trace* 2>/dev/null ;;
    pmc)
      for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR"; do
        t=$(echo $pmc | cut -d' ' -f1)
        (cd /tmp && timeout 600 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$OUT/pmc_$t -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-view-sweep) > $OUT/pmc_$t.log 2>&1
        tail -1 $OUT/pmc_$t.log | cut -c1-200 | tee -a $S
      done
      python tools/pmc_summary.py $OUT 2>&1 | tail -40 | tee -a $S
      python tools/make_traffic.py $OUT "round 6 $TAG" > $OUT/traffic.json 2>/dev/null; head -c 600 $OUT/traffic.json | tee -a $S ;;
    pmc_trained)
      # the same three --pmc passes on the fitted-model-like scene (rows of the settled single-slice frames: <false, 1>)
      for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR"; do
        t=$(echo $pmc | cut -d' ' -f1)
        (cd /tmp && timeout 900 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$OUT/pmct_$t -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-view-sweep --scene trained) > $OUT/pmct_$t.log 2>&1
        tail -1 $OUT/pmct_$t.log | cut -c1-200 | tee -a $S
      done
      mkdir -p $OUT/trained && for t in FETCH_SIZE WRITE_SIZE SQ_WAVES; do rm -rf $OUT/trained/pmc_$t; cp -r $OUT/pmct_$t $OUT/trained/pmc_$t; done
      python tools/pmc_summary.py $OUT/trained 2>&1 | tail -40 | tee $OUT/pmc_trained_summary.txt | tee -a $S ;;
    *) echo "unknown step $step" | tee -a $S ;;
  esac
done
echo "== done ==" | tee -a $S
