"""A/B build of ONE kernel source with extra -D flags: tools/build_variant.py <file.hip> <out.so> [-DX=1 ...]
Compiles csrc/<file.hip> with _build.py's flags + the extra ones and links it with the other objects of the in-tree
build (3dgs-deblur_amd/build/*.o); run the result with GSD_LIB_PATH=<out.so> (tools/gpu_visit.sh abbuild:...)."""
import importlib.util
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
spec = importlib.util.spec_from_file_location("_b", ROOT / "3dgs-deblur_amd" / "_build.py")
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def main():
    srcs, out, extra = sys.argv[1].split("+"), sys.argv[2], sys.argv[3:]      # several files: a.hip+b.hip
    B.build_library(force=False)
    alt = {}
    for src in srcs:
        flags = dict(B.SOURCES)[src]
        obj = Path("/tmp") / (Path(out).stem + "_" + Path(src).stem + ".o")
        subprocess.check_call([B._hipcc(), *B.COMMON, *flags, *extra, "-c", str(B.CSRC / src), "-o", str(obj)])
        alt[src] = str(obj)
    objs = [alt.get(s, str(B.PKG_DIR / "build" / (Path(s).stem + ".o"))) for s, _ in B.SOURCES]
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    print(out)


if __name__ == "__main__":
    main()
