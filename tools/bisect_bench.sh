#!/bin/bash
# On-GPU bisect of a stage time: the same bench.py flags from historical trees copied (without .git, with their built
# library) into _bisect/<name>/ of the snapshot, interleaved with HEAD.  Round 4: found the projection backward's VGPR cliff
# (profiles/r04_bisect_project_bwd_*.log).  Usage on the GPU box: bash tools/bisect_bench.sh
mkdir -p gpurun_out/bisect
for round in 1 2; do
for t in . _bisect/wt_b . _bisect/wt_b; do
  (cd $t && timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary) > gpurun_out/bisect/$(echo $t | tr '/.' '__')_$round.log 2>&1
  python - "$t" gpurun_out/bisect/$(echo $t | tr '/.' '__')_$round.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[1], d['ms_per_step'], d['stage_ms'].get('project_bwd'), d['stage_ms'].get('project_fwd'), d['stage_ms'].get('raster_bwd'))
PY
done; done
