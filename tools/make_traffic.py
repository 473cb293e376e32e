"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE rocprofv3 passes of tools/gpu_visit.sh (step pmc).

HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE doubled per MI355X_MICROARCH.md's HBM section
(gfx950 counts 128-B requests as 64 B); WRITE_SIZE taken as reported.  Usage: make_traffic.py <gpurun_out> <run-tag>
"""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

out, tag = sys.argv[1], sys.argv[2]
KERNELS = ["raster_bwd_sload_kernel", "raster_fwd_sload_kernel", "raster_bwd_kernel_v2", "raster_fwd_slice_kernel",
           "project_fused_fwd_kernel", "project_fused_bwd_sparse_kernel", "slice_counts_exact_kernel",
           "slice_colors_kernel", "emit_open_kernel", "reduce_tuples_wave_kernel", "radix_scatter_kernel",
           "radix_hist_kernel", "slice_records_kernel", "depth_hist_kernel", "depth_find_kernel", "bin_edges_kernel",
           "scan_apply_fused_kernel", "pose_reduce_kernel", "seg_tail_sort_kernel"]
# issue cycles per VALU wave-instruction of the kernel's inner-loop mix (tools/valu_mix.py x tools/valu_bench.hip)
# round 4 (profiles/r04_valu_mix.txt): static mix of the final inner loops, the backward's with all four quadrants hit
MIX = {"raster_fwd_sload_kernel": 785.1 / 221, "raster_bwd_sload_kernel": 1708.0 / 599,
       "raster_fwd_slice_kernel": 345.8 / 90, "raster_bwd_kernel_v2": 485.5 / 125}
# the same mixes in WALL nanoseconds per wave-instruction and SIMD (no clock assumed; an upper bound: the
# micro-benchmark's wall time includes its launch tails)
MIX_NS = {"raster_fwd_sload_kernel": 467.7 / 221, "raster_bwd_sload_kernel": 981.9 / 599,
          "raster_fwd_slice_kernel": 203.9 / 90, "raster_bwd_kernel_v2": 288.2 / 125}

def mean_per_kernel(counter):
    acc = defaultdict(list)
    for f in glob.glob(f"{out}/pmc_{counter}/**/*counter_collection*.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k in KERNELS:
        vals = [v for name, vs in acc.items() if k in name for v in vs]
        if vals:
            res[k] = sum(vals) / len(vals)
    return res


fetch, write = mean_per_kernel("FETCH_SIZE"), mean_per_kernel("WRITE_SIZE")


def sq(counter):
    acc = defaultdict(list)
    for f in glob.glob(f"{out}/pmc_SQ_WAVES/**/*counter_collection*.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k in KERNELS:
        vals = [v for name, vs in acc.items() if k in name for v in vs]
        if vals:
            res[k] = sum(vals) / len(vals)
    return res


insts, wcyc, wait, busy = sq("SQ_INSTS_VALU"), sq("SQ_WAVE_CYCLES"), sq("SQ_WAIT_INST_ANY"), sq("SQ_BUSY_CYCLES")
valu = {"clock_hz": 2.1e9,
        "source": f"rocprofv3 --pmc SQ_* pass of {tag}; mix_cycles_per_inst = issue cycles of the inner loop's static "
                  "instruction mix (tools/valu_mix.py) priced with tools/valu_bench3.hip (profiles/r03_run3_valu_bench3.log); "
                  "clock: s_memtime vs wall in the same micro-benchmark (~2.1 GHz under load)"}
for k in MIX:
    if k in insts:
        valu[k] = {"insts_valu": int(insts[k]), "mix_cycles_per_inst": round(MIX[k], 3),
                   "mix_wall_ns_per_inst": round(MIX_NS[k], 3),
                   "wait_inst_any_frac": round(wait[k] / wcyc[k], 3) if k in wait and k in wcyc else None,
                   # SQ_WAVE_CYCLES counts quad-cycles summed over waves; SQ_BUSY_CYCLES is per shader engine (32)
                   "waves_per_simd": round(wcyc[k] * 4 / (busy[k] / 32.0) / 1024, 2) if k in busy and k in wcyc else None}
def _lib_hash():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_gsd_build", Path(__file__).resolve().parents[1] / "3dgs-deblur_amd" / "_build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash(), mod.kernel_source_hash(), mod.stored_isa_hashes()


doc = {
    "workload": [1000000, 1920, 1080, 5, 1],
    # the kernels these counters were measured on: bench.py only quotes them while the sources still hash to this
    "lib_source_hash": _lib_hash()[0],
    "kernel_source_hash": _lib_hash()[1],       # csrc + flags: what bench.py gates roofline.traffic / .valu on
    # ... or, per kernel, the hash of its ISA (_build.kernel_isa_hashes): a change elsewhere in csrc/ keeps its counters
    "kernel_isa_hash": _lib_hash()[2],
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), {tag}; "
              "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per launch: FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section "
              "(gfx950 counts 128-B requests as 64 B), WRITE_SIZE uncalibrated; tools/make_traffic.py",
    "hbm_bytes_per_step": {k: int((2 * fetch.get(k, 0) + write.get(k, 0)) * 1024) for k in KERNELS if k in fetch or k in write},
    "valu": valu,
}
print(json.dumps(doc, indent=1))
