#!/bin/bash
# per-launch timeline of one step (headline and fitted-model-like scene) under rocprofv3 --kernel-trace
set -u
OUT=gpurun_out/${1:-r3_trace}
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "config5 or native_frame" 2>&1 | tail -3
for scene in survey trained; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof_$scene -o trace -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --scene $scene) > $OUT/rocprof_$scene.log 2>&1
  python tools/trace_step.py $OUT/prof_$scene > $OUT/timeline_$scene.txt 2>&1
  tail -1 $OUT/timeline_$scene.txt
  rm -rf $OUT/prof_$scene
done
