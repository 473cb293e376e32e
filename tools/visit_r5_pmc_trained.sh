# PMC passes (separate runs, --kernel-trace only) of the fitted-model-like scene: HBM bytes and VALU instructions of its compositors
set -u
OUT=gpurun_out/r5_pmc_trained; mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR"; do
  t=$(echo $pmc | cut -d' ' -f1)
  (cd /tmp && timeout 600 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$OUT/pmc_$t -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --scene trained) > $OUT/pmc_$t.log 2>&1
  tail -1 $OUT/pmc_$t.log | cut -c1-200
done
python tools/pmc_summary.py $OUT 2>&1 | grep -E "^---|raster_bwd|raster_fwd|radix_scatter|reduce_tuples|emit_open|slice_counts" | cut -c1-330
