"""developer aid: build libsrbh.so variants that differ in the -D flags of ONE translation unit (default srbh_ptrunk.hip), for
same-box A/B runs (tools/ab_variants.sh).  usage: build_variant.py <tag> [-DNAME=VALUE ...] [--src other.hip]
Writes build/variants/libsrbh_<tag>.so (travels to the GPU box with the snapshot; never loaded by the product)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srbh_amd import _lib
tag, flags, src = sys.argv[1], [a for a in sys.argv[2:] if a.startswith("-")], "srbh_ptrunk.hip"
if "--src" in sys.argv:
    src = sys.argv[sys.argv.index("--src") + 1]
    flags = [f for f in flags if f != "--src"]
_lib.build()
out_dir = os.path.join(_lib.ROOT, "build", "variants")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"{src[:-4]}_{tag}.o")
subprocess.check_call([_lib.HIPCC, *_lib.HIPFLAGS, *flags, "-I", _lib.INCLUDE, "-I", _lib.CSRC, "-c", os.path.join(_lib.CSRC, src), "-o", obj])
objs = [os.path.join(_lib.CSRC, s[:-4] + ".o") if s != src else obj for s in _lib.SOURCES]
so = os.path.join(out_dir, f"libsrbh_{tag}.so")
subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", so])
print(so)
