"""A/B build of the library from PATCHED copies of the kernel sources: tools/ab_patch.py <patch.json> <out.so>
patch.json: [{"file": "raster_bwd.hip", "old": "...", "new": "..."}, ...] (exact string replacements, each must match
once).  The product sources carry no measurement macros; an experiment that needs a different
kernel body lives in tools/patches/ and is applied to a scratch copy of csrc/ here.  Load the result with GSD_LIB_PATH."""
import importlib.util
import json
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("_gsd_build", ROOT / "3dgs-deblur_amd" / "_build.py")
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def main(patch_file: str, out_so: str) -> None:
    patches = json.loads(Path(patch_file).read_text())
    work = Path(tempfile.mkdtemp(prefix="gsd_ab_"))
    pkg = work / "3dgs-deblur_amd"
    shutil.copytree(B.CSRC, pkg / "csrc")
    shutil.copytree(ROOT / "include", work / "include")
    for p in patches:
        f = pkg / "csrc" / p["file"]
        txt = f.read_text()
        assert txt.count(p["old"]) == 1, (p["file"], txt.count(p["old"]), p["old"][:60])
        f.write_text(txt.replace(p["old"], p["new"]))
    objs = []
    for src, extra in B.SOURCES:
        o = work / (src + ".o")
        subprocess.check_call([B._hipcc(), *B.COMMON, *extra, "-c", str(pkg / "csrc" / src), "-o", str(o)])
        objs.append(str(o))
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out_so])
    print(f"built {out_so} from {len(patches)} patch(es)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
