set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "--rs-bands 10 --subposes 1" "--rs-bands 2 --subposes 5" "--gaussians 300000" "--width 3840 --height 2160 --subposes 2" "--gaussians 5000000 --subposes 2"; do
  echo "== bench $cfg" | tee -a gpurun_out/extra.log
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $cfg 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['config']['tile_intersections_per_step'], d['config']['depth_slices'], d['stage_ms'])
    else: print(l[:300])" | tee -a gpurun_out/extra.log
done
