#!/bin/bash
# per-launch durations of the sort micro-benchmark configurations (tools/sort_bench.py) -> gpurun_out/sort_trace.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/sort_trace.txt
for c in "$@"; do
  rm -rf gpurun_out/strc
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/strc -- python tools/sort_bench.py $c 3 > gpurun_out/strc.log 2>&1
  grep "ms per call" gpurun_out/strc.log >> gpurun_out/sort_trace.txt
  python tools/trace_step.py gpurun_out/strc tail:22 >> gpurun_out/sort_trace.txt
done
rm -rf gpurun_out/strc
cat gpurun_out/sort_trace.txt
