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
    ("project.hip", "project_fused_fwd_kernel<16, true>"): (96, False),       # 5 waves
    ("project.hip", "project_fused_bwd_sparse_kernel<16, true>"): (256, True),   # 2 waves: the zero fill needs them
    ("project.hip", "project_needle_hp_kernel"): (168, False),                # 3 waves
    ("binning.hip", "radix_scatter_kernel<unsigned int, 8>"): (168, False),   # 3 waves
    ("binning.hip", "slice_counts_exact_kernel<true>"): (80, False),          # 6 waves
}


def _metadata(asm: str):
    out = {}
    for blk in asm.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = (int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)),
                     int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)))
    return out


@pytest.mark.timeout(900)
def test_hot_kernels_stay_inside_their_register_budgets(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("_gsd_build", ROOT / "3dgs-deblur_amd" / "_build.py")
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    hipcc = B._hipcc()
    if shutil.which(hipcc) is None or shutil.which("c++filt") is None:
        pytest.skip("no hipcc / c++filt on this host")
    flags = dict(B.SOURCES)
    seen = {}
    for src in sorted({s for s, _ in BUDGET}):
        out = tmp_path / (src + ".s")
        subprocess.check_call([hipcc, *B.COMMON, *flags[src], "-S", "--cuda-device-only", str(B.CSRC / src), "-o", str(out)],
                              stderr=subprocess.DEVNULL)
        meta = _metadata(out.read_text())
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
