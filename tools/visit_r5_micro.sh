# packed counts at the field's limit, self-zeroing bin edges, 16-byte combine scale, one-launch flag fill:
# the tests that touch them, then bench + per-launch timeline
set -u
OUT=gpurun_out/r5_micro; mkdir -p $OUT
timeout 800 python -m pytest tests -m gpu -q -p no:cacheprovider -k "sort or compacting or binning or bin_edges or combine or frame or orchestration or knob or golden or full_size_headline_vs or band_aware or speculative or two_step or routes_agree or render_step or rasterize_gaussians_parity" --deselect tests/test_gpu_parity.py::test_baseline_configs_vs_float64_oracle_image_and_per_element_gradients 2>&1 | tail -6
for v in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/h_$v.log 2>&1
  python - $OUT/h_$v.log <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('headline ms', d['ms_per_step'], 'stall', d['host_stall_ms'], d['stage_ms'])
PY
done
bash tools/gpu_visit.sh r5_micro prof 2>&1 | tail -3
tail -3 gpurun_out/r5_micro/timeline.txt
