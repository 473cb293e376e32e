#!/bin/bash
# A/B of compile-time variants of raster.hip: builds /tmp/libgsd_<tag>.so and benches each (interleaved).
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
C="3dgs-deblur_amd/csrc"; B="3dgs-deblur_amd/build"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fvisibility=hidden"
declare -A V=( [m12]="-DGS_FWD_LIVE_MASK=12" [m4]="-DGS_FWD_LIVE_MASK=4" [m0]="" )
for t in "${!V[@]}"; do
  hipcc $FL ${V[$t]} -c $C/raster.hip -o /tmp/raster_$t.o && hipcc --offload-arch=gfx950 -shared -fPIC $B/project.o $B/binning.o /tmp/raster_$t.o $B/raster_bwd.o $B/dp_exchange.o -o /tmp/libgsd_$t.so
done
for rep in 1 2; do
for t in m12 m4 m0; do
  GSD_LIB_PATH=/tmp/libgsd_$t.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$t', d['value'], d['ms_per_step'], 'fwd', d['stage_ms']['raster_fwd'], 'bwd', d['stage_ms']['raster_bwd'])" | tee -a gpurun_out/ab_fwd_build.log
done; done
