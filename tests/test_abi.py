"""C-ABI boundary checks that need no GPU: the library builds for gfx950, loads, and exports
exactly the entry points include/gsdeblur.h declares; the product never touches the oracle and has
no CPU fallback."""
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "3dgs-deblur_amd"


def _declared():
    txt = (ROOT / "include" / "gsdeblur.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(gs):
    lib = gs._lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gsdeblur.h but not exported"
    assert sorted(gs._lib.exported_names()) == names          # the ctypes table covers the header exactly
    assert b"gfx950" in lib.gs_version()


def test_workspace_queries_are_pure_host_calls(gs):
    lib = gs._lib.load()
    assert lib.gs_scan_workspace_bytes(10_000_000) > 0
    a = lib.gs_radix_sort_workspace_bytes(150_000_000, 0, 16)
    b = lib.gs_radix_sort_workspace_bytes(150_000_000, 0, 17)
    assert 0 < a < b < 2 ** 31
    # compacting depth pre-sort: wider digits need a larger histogram; both cover the scan space + the slack that holds
    # the first pass's grand total
    c8 = lib.gs_segmented_sort_compact_workspace_bytes(5_000_000, 1_000_000, 0, 31, 8)
    c11 = lib.gs_segmented_sort_compact_workspace_bytes(5_000_000, 1_000_000, 0, 31, 11)
    assert 0 < c8 < c11
    assert c8 == lib.gs_segmented_sort_workspace_bytes(5_000_000, 1_000_000, 0, 31)
    assert lib.gs_segmented_sort_compact_workspace_bytes(0, 1, 0, 31, 8) == 0


def test_product_never_imports_oracle_or_reference():
    for p in list(PKG.rglob("*.py")) + list(PKG.rglob("*.hip")) + list(PKG.rglob("*.h")) + [ROOT / "gsdeblur_amd.py"]:
        src = p.read_text()
        assert not re.search(r"^\s*(import|from)\s+gs_oracle", src, flags=re.M), p
        assert "oracle/" not in src.replace("oracle/gs_oracle.py::", "") or p.suffix in (".hip", ".h"), p
        assert not re.search(r"open\(.*/root/reference", src), p
        # ... nor the test-only Python orchestration twin of the frame pipeline
        assert not re.search(r"^\s*(import|from)\s+python_frame_path", src, flags=re.M), p


def test_ops_fail_loudly_without_gpu_tensors(gs):
    with pytest.raises(ValueError, match="no CPU fallback"):
        gs.project_gaussians(torch.zeros(4, 3), torch.ones(4, 3), 1.0, torch.ones(4, 4), torch.eye(4), 1, 1, 1, 1,
                             16, 16, 16)
    with pytest.raises(ValueError, match="no CPU fallback"):
        gs.spherical_harmonics(3, torch.zeros(4, 3), torch.zeros(4, 16, 3))
    with pytest.raises(ValueError, match="no CPU fallback"):
        gs.combine_samples(torch.zeros(2, 4, 4, 3), 2.2, 10.0)


def test_missing_library_raises(gs, monkeypatch, tmp_path):
    monkeypatch.setattr(gs._lib, "_lib", None)
    monkeypatch.setattr(gs._lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(gs._lib.HipLibraryError):
        gs._lib.load(build_if_missing=False)


def test_frame_structs_match_the_header_layout(gs, tmp_path):
    """the ctypes mirrors of gs_frame_desc / gs_frame_slice / gs_frame_state / gs_project_inputs (ops.py) against the C header itself: a tiny C
    program compiled with gcc prints sizeof and the offset of every field; a field added on one side only (round 4 added
    shared_list and the combine fields) shows up here, on CPU, instead of as a corrupted frame on the GPU"""
    import ctypes
    import shutil
    import subprocess
    from gsdeblur_amd import ops
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    structs = {"gs_frame_desc": ops._FrameDesc, "gs_frame_slice": ops._FrameSlice, "gs_frame_state": ops._FrameState,
               "gs_project_inputs": ops._ProjectInputs}        # (round 5: what the lazy record projection is called with)
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT / "include" / "gsdeblur.h"}"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(src)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), (cname, got[cname], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)
    # and the header has no field the mirrors lack: count the members between the braces
    hdr = (ROOT / "include" / "gsdeblur.h").read_text()
    for cname, cls in structs.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        n_members = sum(len(decl.split(",")) for decl in body.split(";") if decl.strip())
        assert n_members == len(cls._fields_), (cname, n_members, len(cls._fields_))


def test_ctypes_signatures_match_the_header_prototypes(gs):
    """_lib._SIGS (the ctypes argtypes of every entry point the Python side calls) against the prototypes in
    include/gsdeblur.h: same number of parameters, and the same KIND in every position — pointer / int-like / float /
    long long.  A parameter added to the header and not to the table (round 4 added nine) would otherwise shift every
    later argument by one register on the GPU box, where it shows as a wrong picture or a fault."""
    import ctypes
    from gsdeblur_amd import _lib
    hdr = (ROOT / "include" / "gsdeblur.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", "", hdr)
    protos = {}
    for m in re.finditer(r"\b(gs_\w+)\s*\(([^;{}]*?)\)\s*;", hdr):
        params = [p.strip() for p in m.group(2).split(",")]
        protos[m.group(1)] = [] if params in ([""], ["void"]) else params

    def kind_of_param(p):
        if "*" in p:
            return "ptr"
        if re.search(r"\bdouble\b", p):
            return "double"
        if re.search(r"\bfloat\b", p):
            return "float"
        if re.search(r"\blong long\b", p):
            return "longlong"
        if re.search(r"\b(int|unsigned)\b", p):
            return "int"
        raise AssertionError(f"unclassified parameter {p!r}")

    def kind_of_ctype(t):
        if t in (ctypes.c_void_p, ctypes.c_char_p) or isinstance(t, type(ctypes.POINTER(ctypes.c_int))) and issubclass(t, ctypes._Pointer):
            return "ptr"
        if t is ctypes.c_double:
            return "double"
        if t is ctypes.c_float:
            return "float"
        if t in (ctypes.c_longlong, ctypes.c_ulonglong):
            return "longlong"
        if t in (ctypes.c_int, ctypes.c_uint):
            return "int"
        raise AssertionError(f"unclassified ctype {t!r}")
    assert len(_lib._SIGS) >= 50
    for name, argtypes in _lib._SIGS.items():
        assert name in protos, f"{name} is bound by _lib.py but not declared in the header"
        want = [kind_of_param(p) for p in protos[name]]
        got = [kind_of_ctype(t) for t in argtypes]
        assert got == want, (name, [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w], len(got), len(want))


def test_definitions_match_the_header_prototypes():
    """every `GS_EXPORT ... gs_*(...) {` definition in csrc/*.hip against its prototype in include/gsdeblur.h: parameter
    count and kind (pointer / int / long long / float / double) position by position.  extern "C" symbols carry no
    signature, so a definition that drifts from the header links and loads — and breaks the first C caller that trusts
    the header (round 4 found gs_project_bwd's grad_flags missing from it this way)."""
    def strip(txt):
        return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", txt, flags=re.S))

    def kinds(params):
        out = []
        for p in params:
            if "*" in p:
                out.append("ptr")
            elif re.search(r"\bdouble\b", p):
                out.append("double")
            elif re.search(r"\bfloat\b", p):
                out.append("float")
            elif re.search(r"\blong long\b", p):
                out.append("longlong")
            elif re.search(r"\b(int|unsigned)\b", p):
                out.append("int")
            else:
                raise AssertionError(f"unclassified parameter {p!r}")
        return out

    def split(arglist):
        ps = [p.strip() for p in arglist.split(",")]
        return [] if ps in ([""], ["void"]) else ps
    hdr = strip((ROOT / "include" / "gsdeblur.h").read_text())
    protos = {m.group(1): kinds(split(m.group(2))) for m in re.finditer(r"\b(gs_\w+)\s*\(([^;{}]*?)\)\s*;", hdr)}
    n = 0
    for f in sorted((ROOT / "3dgs-deblur_amd" / "csrc").glob("*.hip")):
        src = strip(f.read_text())
        for m in re.finditer(r"GS_EXPORT\s+[\w\s\*]+?\b(gs_\w+)\s*\(([^{};]*?)\)\s*\{", src):
            name, got = m.group(1), kinds(split(m.group(2)))
            assert name in protos, f"{name} ({f.name}) is exported but not declared in the header"
            assert got == protos[name], (name, f.name, len(got), len(protos[name]),
                                         [(i, a, b) for i, (a, b) in enumerate(zip(got, protos[name])) if a != b])
            n += 1
    assert n == len(protos) and n >= 60, (n, len(protos))


def test_every_call_site_passes_as_many_arguments_as_the_signature_has():
    """static check of the Python side: every `<lib>.gs_*(...)` call in the package, the test twin, bench.py and the tools
    passes exactly as many positional arguments as `_lib._SIGS` (and therefore the header) declares — ctypes would only
    say so at run time, on the GPU box, and only for the call sites a test happens to reach"""
    import ast
    import sys
    sys.path.insert(0, str(ROOT))
    from gsdeblur_amd import _lib
    files = sorted((ROOT / "3dgs-deblur_amd").glob("*.py")) + [ROOT / "tests" / "python_frame_path.py", ROOT / "bench.py",
                                                               ROOT / "__graft_entry__.py"] + \
        sorted((ROOT / "tools").glob("*.py")) + sorted((ROOT / "tests").glob("test_*.py"))
    n = 0
    for f in files:
        tree = ast.parse(f.read_text())
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith("gs_"):
                name = node.func.attr
                if name not in _lib._SIGS or node.keywords or any(isinstance(a, ast.Starred) for a in node.args):
                    continue
                assert len(node.args) == len(_lib._SIGS[name]), (f.name, node.lineno, name, len(node.args),
                                                                 len(_lib._SIGS[name]))
                n += 1
    assert n >= 60, n


def test_the_library_reads_no_environment_and_keeps_no_tunable_statics():
    """include/gsdeblur.h: "No global state; safe to call concurrently on different streams".  VERDICT round 5 weak 11: the
    sort read GSD_SORT_SINGLE_PASS / GSD_COMPACT_PACK through getenv and kept the answer in a static.  Every switch now
    lives in an argument or a descriptor field; the only process-wide state left is the stage profiler of frame.hip, a
    mutex-guarded measurement facility (gs_frame_profile_*)."""
    csrc = ROOT / "3dgs-deblur_amd" / "csrc"
    for f in sorted(csrc.glob("*.hip")) + sorted(csrc.glob("*.h")):
        txt = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", f.read_text(), flags=re.S))
        assert "getenv" not in txt, f.name
        assert not re.search(r"#\s*include\s*<(stdlib\.h|cstdlib)>", txt), f.name
        # file-scope mutable statics: none outside the profiler's
        for m in re.finditer(r"^static\s+(?!inline|constexpr|const\b|__device__|__global__|int\s+run_|void\s|long\s+long\s+\w+\()[^;({]*\b(g_\w+)\b", txt, flags=re.M):
            raise AssertionError(f"{f.name}: mutable file-scope static {m.group(1)}")
