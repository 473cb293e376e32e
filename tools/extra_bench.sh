set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# BASELINE.json configs[1..4] per GPU: 300k/S=5; 1M/10 rolling-shutter bands; 2M blur+RS (one view of the 8-view batch);
# 5M/4K/10 sub-poses
rm -f gpurun_out/extra.log
for cfg in "--gaussians 300000" "--rs-bands 10 --subposes 1" "--gaussians 2000000 --subposes 5 --rs-bands 2" "--gaussians 5000000 --width 3840 --height 2160 --subposes 10"; do
  echo "== bench $cfg" | tee -a gpurun_out/extra.log
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $cfg 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['config']['tile_intersections_per_step'], d['config']['depth_slices'], d['stage_ms'])
    else: print(l[:300])" | tee -a gpurun_out/extra.log
done
