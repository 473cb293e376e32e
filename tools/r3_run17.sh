#!/bin/bash
# per-launch timeline of one headline step with the three-kernel and the single-pass sort forms
set -u
OUT=gpurun_out/r3_run17
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1; do
  (cd /tmp && GSD_SORT_SINGLE_PASS=$m timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof_$m -o trace -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary) > $OUT/rocprof_$m.log 2>&1
  python tools/trace_step.py $OUT/prof_$m > $OUT/timeline_sp$m.txt 2>&1
  tail -1 $OUT/timeline_sp$m.txt
  rm -rf $OUT/prof_$m
done
