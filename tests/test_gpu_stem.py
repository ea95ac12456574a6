"""GPU: the encoder's stem at inference as one libsrbh pass (srbh_stem_conv_eval: conv3x3 stride 2 with the static "same" padding + folded
BatchNorm + SiLU; smp EfficientNetEncoder.forward behind mymodels.py:276) against the stock chain F.conv2d(ZeroPad2d) -> BatchNorm2d(eval) -> x *
sigmoid(x) in float64, at the encoder's own shape and at odd sizes / other pads / channel counts; and the whole encoder with the switch on / off."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,Cin,Cout,H,W,size", [(3, 8, 48, 64, 64, 380), (2, 3, 32, 33, 47, 224), (1, 8, 40, 16, 16, 16), (5, 16, 40, 10, 12, 7)])
def test_stem_kernel_against_the_float64_chain(B, Cin, Cout, H, W, size):
    from srbh_amd import encoders as E
    torch.manual_seed(B + Cin)
    conv = E.SamePadConv2d(Cin, Cout, 3, size, stride=2, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(Cout, momentum=E.BN_MOM, eps=E.BN_EPS).to(DEV).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3); bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn((B, Cin, H, W), device=DEV)
    with torch.no_grad():
        assert E._stem_eval_ok(conv, bn, x)
        y = E._stem_eval(conv, bn, x)
        c64, b64, x64 = conv.double(), bn.double(), x.double()
        z = b64(F.conv2d(c64.static_padding(x64), c64.weight, None, c64.stride))
        want = z * torch.sigmoid(z)
    assert y.shape == want.shape
    rel = float((y.double() - want).norm() / want.norm())
    assert rel <= 2e-6, rel


def test_encoder_eval_with_and_without_the_stem_kernel(monkeypatch):
    from srbh_amd import encoders as E
    torch.manual_seed(0)
    enc = E.EfficientNetEncoder("efficientnet-b4", in_channels=8).to(DEV).eval()
    x = torch.rand((4, 8, 64, 64), device=DEV)
    outs = []
    with torch.no_grad():
        for on in (True, False):
            monkeypatch.setattr(E, "STEM_EVAL", on)
            outs.append([f.clone() for f in enc(x)])
    for a, b in zip(*outs):
        assert float((a - b).norm() / b.norm().clamp_min(1e-30)) <= 1e-5
