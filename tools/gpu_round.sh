#!/bin/bash
# One GPU-box visit: parity tests, smoke, micro-benchmarks, bench, rocprof kernel trace + PMC passes.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh [quick|full|pmc]
set -u
MODE=${1:-full}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== build ==" | tee $OUT/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2 | tee -a $OUT/summary.log
if [ "$MODE" != "pmc" ]; then
echo "== pytest -m gpu ==" | tee -a $OUT/summary.log
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
grep -E "ulp diffs|^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -30 | tee -a $OUT/summary.log
echo "== smoke ==" | tee -a $OUT/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $OUT/summary.log
fi
if [ "$MODE" = "full" ]; then
  echo "== atomic microbench ==" | tee -a $OUT/summary.log
  (hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -Wno-unused-value tools/atomic_bench.hip -o /tmp/atomic_bench && timeout 120 /tmp/atomic_bench) 2>&1 | tail -12 | tee -a $OUT/summary.log
fi
if [ "$MODE" = "full" ]; then
  echo "== dp exchange, device side ==" | tee -a $OUT/summary.log
  timeout 120 python tools/dp_bench.py 2>&1 | tail -2 | tee -a $OUT/summary.log
  echo "== dp exchange, device side, fitted-model-like density (33 % of the rows) ==" | tee -a $OUT/summary.log
  timeout 120 python tools/dp_bench.py 1000000 330000 8 2>&1 | tail -2 | tee -a $OUT/summary.log
fi
echo "== bench x2 (no cpu baseline) ==" | tee -a $OUT/summary.log
for v in 1 2; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('run $v:', d['value'], d['ms_per_step'], d['stage_ms'], d['config'].get('depth_slices'))" | tee -a $OUT/summary.log
done
echo "== bench default ==" | tee -a $OUT/summary.log
timeout 600 python bench.py > $OUT/bench.log 2>&1
tail -2 $OUT/bench.log | cut -c1-2500 | tee -a $OUT/summary.log
if [ "$MODE" != "quick" ]; then
  echo "== rocprofv3 kernel trace ==" | tee -a $OUT/summary.log
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary) > $OUT/rocprof.log 2>&1
  grep -o '{"metric.*' $OUT/rocprof.log | cut -c1-400 | tee -a $OUT/summary.log
  for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -22 $f | cut -c1-220 | tee -a $OUT/summary.log; done
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    echo "== rocprofv3 pmc $tag ==" | tee -a $OUT/summary.log
    (cd /tmp && timeout 600 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$OUT/pmc_$tag -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary) > $OUT/pmc_$tag.log 2>&1
    tail -2 $OUT/pmc_$tag.log | cut -c1-300 | tee -a $OUT/summary.log
    find $OUT/pmc_$tag -name "*.csv" | head -5 | tee -a $OUT/summary.log
  done
  python tools/pmc_summary.py $OUT 2>&1 | tail -40 | tee -a $OUT/summary.log
  python tools/make_traffic.py $OUT "round 3 ${TAG:-final}" > $OUT/traffic.json 2>/dev/null; head -c 600 $OUT/traffic.json | tee -a $OUT/summary.log
  echo "== rocprofv3 kernel trace, fitted-model-like scene ==" | tee -a $OUT/summary.log
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_trained -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --scene trained) > $OUT/rocprof_trained.log 2>&1
  for f in $(find $OUT/prof_trained -name "*kernel_stats*.csv" | head -1); do head -16 $f | cut -c1-200 | tee -a $OUT/summary.log; cp $f $OUT/kernel_stats_trained.csv; done
  for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/kernel_stats.csv; done
  python tools/trace_step.py $OUT/prof > $OUT/timeline.txt 2>/dev/null; tail -1 $OUT/timeline.txt | tee -a $OUT/summary.log
  rm -rf $OUT/prof_trained/*/*kernel_trace* $OUT/prof/*/*kernel_trace* 2>/dev/null
fi
echo "== done ==" | tee -a $OUT/summary.log
