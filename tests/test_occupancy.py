"""Register budgets of the hot kernels, checked on the build host (hipcc cross-compiles gfx950 without a GPU).

Round 4 lost 30 us per step for a day because one kernel went from 256 to 258 VGPRs (two waves per SIMD -> one): a
number nobody looks at until a stage time moves.  This test compiles the kernel sources to assembly with the
library's own flags and holds every hot kernel to the occupancy line DESIGN.md section 4 quotes for it."""
import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

# (source, kernel name fragment as it appears demangled) -> (max VGPRs, may spill?)
BUDGET = {
    ("raster.hip", "raster_fwd_sload_kernel<false>"): (72, False),            # 7 waves per SIMD
    ("raster_bwd.hip", "raster_bwd_sload_kernel<false, 1>"): (96, False),     # 5 waves (launch bound)
    ("raster_bwd.hip", "raster_bwd_sload_kernel<true, 1>"): (96, False),
    ("raster_rs.hip", "raster_fwd_rs_kernel<false>"): (72, False),            # round 6: the pixel-velocity compositors
    ("raster_rs.hip", "raster_bwd_rs_kernel<false>"): (96, False),            # in the current kernel generation (was 140)
    ("raster_rs.hip", "raster_bwd_rs_kernel<true>"): (104, False),
    ("project.hip", "project_fused_fwd_kernel<16, true>"): (96, False),       # 5 waves
    ("project.hip", "project_fused_bwd_sparse_kernel<16, true>"): (256, True),   # 2 waves: the zero fill needs them
    ("project.hip", "project_needle_hp_kernel"): (168, False),                # 3 waves
    ("binning.hip", "radix_scatter_kernel<unsigned int, 8, 0>"): (168, False),   # 3 waves (the tile sort's passes)
    ("binning.hip", "radix_scatter_kernel<unsigned int, 8, 1>"): (168, False),   # depth pre-sort: counts packed ...
    ("binning.hip", "radix_scatter_kernel<unsigned int, 8, 2>"): (168, False),   # ... and unpacked
    ("binning.hip", "seg_tail_sort_kernel"): (128, True),                     # 1024 threads: 24 keys + payloads per thread
    ("binning.hip", "slice_counts_exact_kernel<true, false>"): (80, False),   # 6 waves
    ("binning.hip", "slice_counts_exact_kernel<true, true>"): (96, False),    # the swept form (pixel-velocity lists)
}


def _metadata(asm: str):
    out = {}
    for blk in asm.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = (int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)),
                     int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)))
    return out


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    """source name -> assembly text, compiled with the library's own flags"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_gsd_build", ROOT / "3dgs-deblur_amd" / "_build.py")
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    hipcc = B._hipcc()
    if shutil.which(hipcc) is None or shutil.which("c++filt") is None:
        pytest.skip("no hipcc / c++filt on this host")
    flags = dict(B.SOURCES)
    d = tmp_path_factory.mktemp("asm")
    out = {}
    for src in sorted({s for s, _ in BUDGET}):
        f = d / (src + ".s")
        subprocess.check_call([hipcc, *B.COMMON, *flags[src], "-S", "--cuda-device-only", str(B.CSRC / src), "-o", str(f)],
                              stderr=subprocess.DEVNULL)
        out[src] = f.read_text()
    return out


@pytest.mark.timeout(900)
def test_hot_kernels_stay_inside_their_register_budgets(asm):
    seen = {}
    for src in sorted({s for s, _ in BUDGET}):
        meta = _metadata(asm[src])
        names = list(meta)
        demangled = subprocess.check_output(["c++filt"], input="\n".join(names), text=True).splitlines()
        for mangled, dem in zip(names, demangled):
            seen[(src, dem)] = meta[mangled]
    report = []
    for (src, frag), (max_vgpr, may_spill) in BUDGET.items():
        hits = [(dem, v) for (s, dem), v in seen.items() if s == src and frag in dem]
        assert hits, f"{frag} not found in {src}"
        for dem, (vgpr, spill) in hits:
            report.append(f"{frag}: {vgpr} VGPRs (budget {max_vgpr}), {spill} spilled")
            assert vgpr <= max_vgpr, (frag, vgpr, max_vgpr)
            assert may_spill or spill == 0, (frag, spill)
    print("\n".join(report))


@pytest.mark.timeout(900)
def test_compositor_inner_loops_keep_their_instruction_counts(asm):
    """the VALU instructions of one trip of the compositors' hot loops (a group of four list entries; tools/valu_mix.py's
    loop finder and profiles/r04_valu_mix.txt's numbers: forward 221, backward 599): round 4's work on them was removing
    instructions the compiler adds when a branch merges values (8-30 v_mov per entry at one point) — a change that brings
    them back shows here, not in a stage time a visit later"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("valu_mix", ROOT / "tools" / "valu_mix.py")
    VM = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(VM)
    limits = {("raster.hip", "_ZN2gs23raster_fwd_sload_kernelILb0EEE"): 232,
              ("raster_bwd.hip", "_ZN2gs23raster_bwd_sload_kernelILb0ELi1EEE"): 625}
    for (src, prefix), limit in limits.items():
        m = re.search(r"^(%s\w*):[^\n]*\n(.*?)\.Lfunc_end" % prefix, asm[src], flags=re.S | re.M)
        assert m, prefix
        lines, ls = VM.loops(m.group(2))
        ls = [r for r in ls if r[1] - r[0] < 1500]

        def score(r):
            return sum(1 for ln in lines[r[0]:r[1] + 1] if ln.strip().startswith(("v_exp", "v_rcp")))
        a, b = max(ls, key=lambda r: (score(r), -(r[1] - r[0])))
        n_valu = sum(1 for ln in lines[a:b + 1] if ln.strip() and not ln.strip().startswith((";", "."))
                     and VM.classify(ln.strip()) is not None)
        print(f"{prefix}: {n_valu} VALU instructions per group of four entries (limit {limit})")
        assert n_valu <= limit, (prefix, n_valu, limit)
