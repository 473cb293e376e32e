#!/bin/bash
# One GPU-box visit: parity tests, smoke, micro-benchmarks, bench, rocprof kernel trace.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh [quick|full]
set -u
MODE=${1:-full}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== build ==" | tee $OUT/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2 | tee -a $OUT/summary.log
echo "== pytest -m gpu ==" | tee -a $OUT/summary.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
tail -40 $OUT/pytest_gpu.log | tee -a $OUT/summary.log
echo "== smoke ==" | tee -a $OUT/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee -a $OUT/summary.log
if [ "$MODE" = "full" ]; then
  echo "== atomic microbench ==" | tee -a $OUT/summary.log
  (hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -Wno-unused-value tools/atomic_bench.hip -o /tmp/atomic_bench && timeout 120 /tmp/atomic_bench) 2>&1 | tail -8 | tee -a $OUT/summary.log
fi
echo "== bench small (100k, 1 step) ==" | tee -a $OUT/summary.log
timeout 300 python bench.py --gaussians 100000 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | tee -a $OUT/summary.log
echo "== bench default ==" | tee -a $OUT/summary.log
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.log 2>&1
tail -3 $OUT/bench.log | tee -a $OUT/summary.log
if [ "$MODE" = "full" ]; then
  echo "== rocprofv3 kernel trace ==" | tee -a $OUT/summary.log
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline) > $OUT/rocprof.log 2>&1
  tail -3 $OUT/rocprof.log | tee -a $OUT/summary.log
  find $OUT/prof -name "*stats*" | head | tee -a $OUT/summary.log
  for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -25 $f | cut -c1-200 | tee -a $OUT/summary.log; done
fi
echo "== done ==" | tee -a $OUT/summary.log
