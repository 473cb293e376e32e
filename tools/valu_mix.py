"""Static VALU instruction mix of the compositor kernels' inner loops, priced with the issue rates measured by
tools/valu_bench.hip (profiles/r02_run1_valu_bench.log, 4 waves per SIMD, cycles per wave-instruction per SIMD).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps -c 3dgs-deblur_amd/csrc/raster.hip        (and raster_bwd.hip
  with -fno-slp-vectorize), then:  python tools/valu_mix.py <dir with the .s files>

For every kernel: the instructions between the innermost loop header and its back edge, by class, and the issue
cycles one loop iteration costs a SIMD if VALU issue were the only limit.
"""
import re
import sys
from collections import Counter
from pathlib import Path

COST = {  # measured, w/SIMD=4 column.  Round 3 re-measured them on pinned registers (tools/valu_bench3.hip,
    # profiles/r03_run3_valu_bench3.log): two-source mul / add / mov / integer ops 2.17, fma 2.08 when its second and
    # third source sit in registers of different parity (3.03 otherwise: priced at the mean), anything with an SGPR
    # operand / v_min / v_max / v_cndmask / DPP 3.2-3.6, v_cmp 3.2-3.4 (round 2 said 4.0), v_pk_* 3.49, v_exp / v_rcp 6.45
    "mul/add/fmac, VGPR operands": 2.17, "fma (VOP3), VGPR operands": 2.55, "fma/mul/add, SGPR operand": 3.25,
    "v_pk_*_f32": 3.49, "v_min/max/cndmask/dpp": 3.30, "v_mov/int/other": 2.17, "v_cmp": 3.30, "v_exp/v_rcp": 6.45,
    "v_readlane": 7.89}
# the same classes in WALL nanoseconds per wave-instruction and SIMD (the bench's hipEvent column: no clock assumed)
NS = {"mul/add/fmac, VGPR operands": 1.17, "fma (VOP3), VGPR operands": 1.58, "fma/mul/add, SGPR operand": 1.95,
      "v_pk_*_f32": 2.15, "v_min/max/cndmask/dpp": 1.93, "v_mov/int/other": 1.15, "v_cmp": 1.92, "v_exp/v_rcp": 3.8,
      "v_readlane": 4.6}


def classify(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        return None
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return "v_readlane"
    if op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_log") or op.startswith("v_sqrt"):
        return "v_exp/v_rcp"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("v_pk_"):
        return "v_pk_*_f32"
    args = line.split(None, 1)[1] if len(line.split(None, 1)) > 1 else ""
    if re.match(r"v_(fma|fmac|mul|add|sub|mac|mad)_f32", op):
        srcs = args.split(",")[1:]
        if any(re.search(r"\bs\d+|\bs\[", a) for a in srcs):
            return "fma/mul/add, SGPR operand"
        if "dpp" in line or "quad_perm" in line or "row_" in line:
            return "v_min/max/cndmask/dpp"
        return "fma (VOP3), VGPR operands" if op.startswith("v_fma_") else "mul/add/fmac, VGPR operands"
    if re.match(r"v_(min|max|med3|cndmask)", op) or "dpp" in line or "quad_perm" in line or "row_" in line:
        return "v_min/max/cndmask/dpp"
    return "v_mov/int/other"


def loops(body):
    """(start, end) line ranges of loops: a label up to the LAST later branch back to it"""
    lines = body.splitlines()
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    out = {}
    for j, ln in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)\b", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < j:
            out[labels[m.group(1)]] = j
    return lines, sorted(out.items())


def main(d):
    for f in sorted(Path(d).glob("*gfx950.s")):
        txt = f.read_text()
        for m in re.finditer(r"^(_ZN2gs\d+raster_\w+?kernel\w*):[^\n]*\n(.*?)\.Lfunc_end", txt, flags=re.S | re.M):
            name, body = m.group(1), m.group(2)
            lines, ls = loops(body)
            if not ls:
                continue
            # the hot loop: the one holding the most transcendental instructions per line span under 1500 lines
            def score(r):
                return sum(1 for ln in lines[r[0]:r[1] + 1] if ln.strip().startswith(("v_exp", "v_rcp")))
            ls = [r for r in ls if r[1] - r[0] < 1500]
            a, b = max(ls, key=lambda r: (score(r), -(r[1] - r[0])))
            cnt = Counter()
            salu = lds = vmem = smem = 0
            for ln in lines[a:b + 1]:
                t = ln.strip()
                if not t or t.startswith(";") or t.startswith("."):
                    continue
                c = classify(t)
                if c:
                    cnt[c] += 1
                elif t.startswith("s_load"):
                    smem += 1
                elif t.startswith("s_"):
                    salu += 1
                elif t.startswith("ds_"):
                    lds += 1
                elif t.startswith("global_") or t.startswith("buffer_"):
                    vmem += 1
            total = sum(cnt.values())
            cyc = sum(COST[k] * v for k, v in cnt.items())
            ns = sum(NS[k] * v for k, v in cnt.items())
            short = re.sub(r"^_ZN2gs\d+", "", name)[:44]
            print(f"{short:46s} loop lines {b - a + 1:5d}  VALU {total:4d}  issue cycles {cyc:7.1f}  "
                  f"wall ns {ns:7.1f}  SALU {salu:4d} SMEM {smem:3d} LDS {lds:3d} VMEM {vmem:3d}")
            for k in COST:
                if cnt[k]:
                    print(f"    {k:32s} {cnt[k]:4d} x {COST[k]:.2f} cyc = {cnt[k] * COST[k]:7.1f}    x {NS[k]:.2f} ns = {cnt[k] * NS[k]:7.1f}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
