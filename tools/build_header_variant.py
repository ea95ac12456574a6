"""developer aid: build build/variants/libsrbh_<tag>.so with ONE translation unit compiled against an alternative copy of a header it includes
(experiments that are a source edit, not a -D knob: the candidate lives outside the tree until it wins).
usage: build_header_variant.py <tag> <unit.hip> <header name in csrc> <path of the candidate header>"""
import os, shutil, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srbh_amd import _lib
tag, unit, hname, cand = sys.argv[1:5]
_lib.build()
vdir = os.path.join(_lib.ROOT, "build", "variants")
sdir = os.path.join(vdir, "src_" + tag)
os.makedirs(sdir, exist_ok=True)
shutil.copy(os.path.join(_lib.CSRC, unit), os.path.join(sdir, unit))      # (quote-includes resolve next to the including file first)
shutil.copy(cand, os.path.join(sdir, hname))
obj = os.path.join(vdir, f"{unit[:-4]}_{tag}.o")
subprocess.check_call([_lib.HIPCC, *_lib.HIPFLAGS, "-I", _lib.INCLUDE, "-I", _lib.CSRC, "-c", os.path.join(sdir, unit), "-o", obj])
objs = [os.path.join(_lib.CSRC, s[:-4] + ".o") if s != unit else obj for s in _lib.SOURCES]
so = os.path.join(vdir, f"libsrbh_{tag}.so")
subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", so])
print(so)
