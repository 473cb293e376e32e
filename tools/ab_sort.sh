#!/bin/bash
# A/B of compile-time variants of binning.hip (keys per sort block): builds /tmp/libgsd_<tag>.so, benches each.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
C="3dgs-deblur_amd/csrc"; B="3dgs-deblur_amd/build"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fvisibility=hidden"
declare -A V=( [r16]="" [r8]="-DGS_SORT_ROUNDS_U32=8" [r4]="-DGS_SORT_ROUNDS_U32=4" )
for t in "${!V[@]}"; do
  hipcc $FL ${V[$t]} -c $C/binning.hip -o /tmp/binning_$t.o && hipcc --offload-arch=gfx950 -shared -fPIC $B/project.o /tmp/binning_$t.o $B/raster.o $B/raster_bwd.o $B/dp_exchange.o -o /tmp/libgsd_$t.so
done
for rep in 1 2; do
for t in r16 r8 r4; do
  GSD_LIB_PATH=/tmp/libgsd_$t.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$t', d['value'], d['ms_per_step'], 'depth_sort', d['stage_ms']['depth_sort'], 'tile_sort', d['stage_ms']['tile_sort'])" | tee -a gpurun_out/ab_sort.log
done; done
