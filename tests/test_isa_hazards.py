"""The kernels that carry inline asm (LDS-DMA with explicit EXEC/M0, scalar-base stores) are compiled to ISA text and scanned
for the gfx950 hazards hipcc cannot see through an asm statement (tools/hazcheck.py): a VALU-written SGPR read by a VMEM
instruction within 5 wait states, a freshly written VGPR read by v_readlane/v_readfirstlane, vector registers handed to
scalar instructions.  Both were real bugs during development (wrong addresses / masks, silently)."""
import os, shutil, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "super-resolution-building-height-estimation_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
@pytest.mark.parametrize("src", ["srbh_ptrunk.hip", "srbh_ptail.hip", "srbh_conv3x3.hip"])
def test_no_unseen_hazards_around_inline_asm(src, tmp_path):
    out = str(tmp_path / (src + ".s"))
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hazcheck.py"), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().splitlines()[-1] == "hazards: 0", r.stdout[-3000:]
