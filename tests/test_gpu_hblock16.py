"""GPU: a plain BasicBlock of the fp16 inference chain as ONE libsrbh pass (srbh_hblock16_eval, csrc/srbh_hblock16_kernel.h; reference:
SR/HRfuse.py:142-159 in eval mode) against (1) the two-launch chain it replaces -- same operand rounding and epilogues; since the kernel packs
two taps into one 16x16x32 matrix instruction the 144 products of an output are summed in another order, so the results agree up to the last
fp16 bit on a small fraction of the elements (they were bit-identical with one tap per instruction) -- at every tile position of an image
(corners, edges, interior), fp16 and fp32 outputs -- and (2) the float64 torch graph of the block on the fp16-rounded operands."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _block(seed):
    from srbh_amd import hrfuse as H
    torch.manual_seed(seed)
    blk = H.BasicBlock(16, 16)
    with torch.no_grad():
        for bn in (blk.bn1, blk.bn2):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
            bn.running_mean.uniform_(-0.2, 0.2)
            bn.running_var.uniform_(0.5, 1.5)
    return blk.to(DEV).eval()


@pytest.mark.parametrize("B,Hh,Ww", [(1, 4, 64), (2, 8, 64), (1, 12, 192), (3, 64, 128), (2, 256, 256)])
@pytest.mark.parametrize("out_h16", [True, False])
def test_fused_block_equals_the_two_launch_chain(B, Hh, Ww, out_h16):
    from srbh_amd import hrfuse as H
    blk = _block(3)
    g = torch.Generator().manual_seed(B * 1000 + Hh + Ww)
    x = (torch.randn((B, 16, Hh, Ww), generator=g) * 0.7).to(DEV).to(torch.float16).contiguous(memory_format=torch.channels_last)
    outs = []
    with torch.no_grad(), H.head_precision("f16"):
        for fused in (True, False):
            H.HBLOCK16 = fused
            try:
                from srbh_amd import _lib
                _lib.path_counters(reset=True)
                y = blk.forward_nhwc([x], out_h16=out_h16)
                torch.cuda.synchronize()
                assert _lib.path_counters()["hblock16"] == (1 if fused else 0)
            finally:
                H.HBLOCK16 = True
            assert y.dtype == (torch.float16 if out_h16 else torch.float32)
            outs.append(y.float().cpu())
    assert bool(torch.isfinite(outs[0]).all())
    d = (outs[0] - outs[1]).abs()
    # a different summation order moves an fp32 sum by ~1e-7; where that crosses an fp16 rounding boundary (of the intermediate a1 or of the
    # output) an element moves by one fp16 step: 2^-10 relative, allow two
    assert bool((d <= 2.0 ** -9 * outs[1].abs().clamp_min(1.0)).all()), float(d.max())
    assert float(d.norm() / outs[1].norm()) <= 1e-4
    if out_h16:
        assert float((d > 0).float().mean()) <= 0.02, float((d > 0).float().mean())


def test_fused_block_against_the_float64_graph():
    from srbh_amd import hrfuse as H
    blk = _block(5)
    g = torch.Generator().manual_seed(11)
    x = (torch.randn((2, 16, 20, 128), generator=g) * 0.7).to(torch.float16)
    with torch.no_grad(), H.head_precision("f16"):
        y = blk.forward_nhwc([x.to(DEV).contiguous(memory_format=torch.channels_last)], out_h16=False).cpu().double()
    b64 = _block(5).cpu().double()
    with torch.no_grad():
        w1 = b64.conv1.weight.half().double()
        w2 = b64.conv2.weight.half().double()
        xd = x.double()
        a1 = F.relu(b64.bn1(F.conv2d(xd, w1, padding=1))).half().double()
        want = F.relu(b64.bn2(F.conv2d(a1, w2, padding=1)) + xd)
    rel = float((y - want).norm() / want.norm())
    assert rel <= 2e-5, rel          # (same rounded operands: what is left is summation order and the fp32 affine)
