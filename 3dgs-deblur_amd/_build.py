"""Builds the C-ABI HIP library (libgsdeblur_hip.so) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs on the CPU-only build box; the
resulting .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libgsdeblur_hip.so"

# (source, extra flags).  project.hip must not contract a*b+c into fma: its integer
# outputs (radii, tile bounds, depth key bits) are compared bit-for-bit with the oracle.
SOURCES = [
    ("project.hip", ["-ffp-contract=off"]),
    ("binning.hip", []),
    ("raster.hip", []),
    # backward compositor: automatic SLP packing into v_pk_*_f32 costs v_mov shuffles and 29 VGPRs on gfx950
    # (A/B: 1.27 -> 1.09 ms); the packing is done by hand in the source instead (GS_BWD_PK)
    ("raster_bwd.hip", ["-fno-slp-vectorize"]),
    ("raster_rs.hip", []),
    # x*scale + y must round twice, like the torch ops it replaces (tests compare bit for bit)
    ("dp_exchange.hip", ["-ffp-contract=off"]),
    ("train.hip", []),
    ("frame.hip", []),           # host-side frame orchestration (no kernels of its own)
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fvisibility=hidden"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


HASH_PATH = PKG_DIR / "libgsdeblur_hip.so.srchash"


def source_hash() -> str:
    """content hash of every kernel source, header and of this file (the flags): what the binary was built from"""
    import hashlib
    h = hashlib.sha256()
    for f in (sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "gsdeblur.h"]
              + [Path(__file__)]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def kernel_source_hash() -> str:
    """content hash of what the DEVICE code is compiled from: csrc/*.hip, csrc/*.h and the compiler flags — not the public
    header's prototypes and comments, not this file's text.  Gates the PMC-derived numbers of profiles/traffic.json
    (bench.py roofline.traffic / roofline.valu): they stay valid across a header-only change."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.h")):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    h.update(repr((SOURCES, COMMON)).encode())
    return h.hexdigest()


# ---- per-kernel ISA hashes ------------------------------------------------------------------------------------------
# profiles/traffic.json holds PMC counters of single kernels.  kernel_source_hash() invalidates ALL of them whenever any
# kernel source changes; a counter of kernel K only depends on K's machine code (and on K's inputs), so the file also
# records the hash of K's gfx950 ISA, and bench.py accepts the counters of a kernel whose ISA is still the measured one.
ISA_KERNELS = {"raster.hip": ("raster_fwd_sload_kernel",), "raster_bwd.hip": ("raster_bwd_sload_kernel",)}
ISA_HASH_PATH = PKG_DIR / "libgsdeblur_hip.so.isahash"


def kernel_isa_hashes(csrc: Path = CSRC) -> dict:
    """{kernel name fragment: sha256 over the ISA text of every instantiation of that kernel} for ISA_KERNELS, compiled
    from `csrc` with the library's flags (-S, device side only); comments dropped, basic-block labels renumbered"""
    import hashlib
    import re
    import tempfile
    flags = dict(SOURCES)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for src, frags in ISA_KERNELS.items():
            f = Path(d) / (src + ".s")
            subprocess.check_call([_hipcc(), *COMMON, *flags[src], "-S", "--cuda-device-only", str(csrc / src), "-o", str(f)],
                                  stderr=subprocess.DEVNULL)
            lines = f.read_text().split("\n")
            bodies = {}
            name = None
            for ln in lines:
                m = re.match(r"^(_Z\w+):", ln)
                if m:
                    name, bodies[m.group(1)] = m.group(1), []
                    continue
                if name is None:
                    continue
                if re.match(r"^\.Lfunc_end\d+:", ln):
                    name = None
                    continue
                t = ln.split(";")[0].strip()
                if t:
                    bodies[name].append(re.sub(r"\.LBB\d+_", ".LBB_", t))
            for frag in frags:
                h = hashlib.sha256()
                hits = sorted(k for k in bodies if frag in k)
                if not hits:
                    raise RuntimeError(f"{frag}: no kernel of that name in {src}")
                for k in hits:
                    h.update(k.encode())
                    h.update("\n".join(bodies[k]).encode())
                out[frag] = h.hexdigest()
    return out


def stored_isa_hashes() -> dict:
    """the ISA hashes written next to the in-tree library by build_library() ({} when absent or stale)"""
    import json
    try:
        d = json.loads(ISA_HASH_PATH.read_text())
        return d["kernels"] if d.get("source_hash") == source_hash() else {}
    except Exception:
        return {}


def is_current() -> bool:
    """True when the in-tree library exists and was built from exactly the sources that are on disk now (content,
    not mtime: a repository snapshot copied to another box keeps the bytes but not necessarily the timestamps)"""
    return LIB_PATH.exists() and HASH_PATH.exists() and HASH_PATH.read_text().strip() == source_hash()


def build_library(force: bool = False, verbose: bool = False) -> Path:
    if not force and is_current():
        return LIB_PATH
    force = force or LIB_PATH.exists()      # a stale binary: rebuild every object (mtimes cannot be trusted)
    headers = sorted(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "gsdeblur.h"]
    objs = []
    hipcc = _hipcc()
    build_dir = PKG_DIR / "build"
    build_dir.mkdir(exist_ok=True)
    for src, extra in SOURCES:
        s = CSRC / src
        o = build_dir / (s.stem + ".o")
        if force or _stale(o, [s, *headers, Path(__file__)]):     # the flags live in this file
            cmd = [hipcc, *COMMON, *extra, "-c", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB_PATH)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    HASH_PATH.write_text(source_hash() + "\n")
    try:
        import json
        ISA_HASH_PATH.write_text(json.dumps({"source_hash": source_hash(), "kernels": kernel_isa_hashes()}) + "\n")
    except Exception:                       # the counters of profiles/traffic.json then fall back to kernel_source_hash
        ISA_HASH_PATH.unlink(missing_ok=True)
    return LIB_PATH


# ---- TEST infrastructure: the round-1 compositors (v_readlane kernels; the "plain path" of the equivalence tests and the
# lane-utilisation counters) are compiled out of the product library (GS_ROUND1_KERNELS) and live in a library of their
# own under tests/, built from the same two sources with -DGS_ROUND1_KERNELS=1 (tests/python_frame_path.py loads it).
ROUND1_LIB_PATH = PKG_DIR.parent / "tests" / "libgsdeblur_round1.so"
ROUND1_HASH_PATH = PKG_DIR.parent / "tests" / "libgsdeblur_round1.so.srchash"


def build_round1_library(force: bool = False, verbose: bool = False) -> Path:
    if not force and ROUND1_LIB_PATH.exists() and ROUND1_HASH_PATH.exists() and \
            ROUND1_HASH_PATH.read_text().strip() == source_hash():
        return ROUND1_LIB_PATH
    hipcc = _hipcc()
    build_dir = PKG_DIR / "build"
    build_dir.mkdir(exist_ok=True)
    flags = dict(SOURCES)
    objs = []
    for src in ("raster.hip", "raster_bwd.hip"):
        o = build_dir / (Path(src).stem + "_round1.o")
        cmd = [hipcc, *COMMON, *flags[src], "-DGS_ROUND1_KERNELS=1", "-c", str(CSRC / src), "-o", str(o)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(ROUND1_LIB_PATH)]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    ROUND1_HASH_PATH.write_text(source_hash() + "\n")
    return ROUND1_LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_round1_library(force=True, verbose=True))
