"""Build variant libraries for same-box A/B runs:  python tools/mkvariants.py SPEC.py

SPEC.py defines VARIANTS = {name: (file.hip, [(old, new), ...])}: every `old` must occur exactly once in that source file.  Each
variant is compiled from the patched source with the Makefile's flags and linked with the tree's other objects into
s3gaussian_amd/lib/variants/libs3g_<name>.so (git-ignored; travels with gpurun).  On the GPU box:

    for v in a b c; do S3G_LIB_PATH=$PWD/s3gaussian_amd/lib/variants/libs3g_$v.so python tools/diag_split.py; done

(the ABI must be unchanged; `make -C s3gaussian_amd/csrc` first, so that the other objects are current).  This is how the
packed-VALU -> LDS-store hazard of the split inference kernel was isolated in three 20-second GPU calls (profiles/r03_split_hazard.jsonl)."""
import glob
import os
import runpy
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "s3gaussian_amd", "csrc")
LIB = os.path.join(ROOT, "s3gaussian_amd", "lib")
OUT = os.path.join(LIB, "variants")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics"]


def build(name, hip, edits):
    src = open(os.path.join(SRC, hip)).read()
    for old, new in edits:
        if src.count(old) != 1:
            raise SystemExit(f"{name}: pattern occurs {src.count(old)} times in {hip}: {old[:70]!r}")
        src = src.replace(old, new)
    tmp_src = os.path.join(SRC, f"_variant_{name}.hip")      # next to the real sources: relative #includes keep working
    obj = os.path.join(OUT, f"{name}.o")
    try:
        open(tmp_src, "w").write(src)
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-c", tmp_src, "-o", obj])
    finally:
        if os.path.exists(tmp_src):
            os.remove(tmp_src)
    base = os.path.splitext(hip)[0]
    objs = [o for o in glob.glob(os.path.join(LIB, "*.o")) if os.path.basename(o) != base + ".o"] + [obj]
    so = os.path.join(OUT, f"libs3g_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", so])
    os.remove(obj)
    return so


def main():
    if len(sys.argv) != 2:
        raise SystemExit(__doc__)
    variants = runpy.run_path(sys.argv[1])["VARIANTS"]
    os.makedirs(OUT, exist_ok=True)
    for name, (hip, edits) in variants.items():
        print(build(name, hip, edits))


if __name__ == "__main__":
    main()
