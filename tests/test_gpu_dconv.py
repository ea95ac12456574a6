"""GPU: the U-Net decoders' 3x3 convolutions on libsrbh (csrc/srbh_dconv.hip; reference: smp UnetDecoder blocks built at
mymodels.py:245-258, called at :279 / :287).  Forward (fp16 operands), data gradient (bf16) and weight gradient (bf16) against float64
stock convolutions of the SAME rounded operands (what remains is fp32 summation order), and against the unrounded ones at the 16-bit
operand level; every decoder shape of the model, ragged batch sizes, odd channel counts."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEV = "cuda:0"

# (Cin, Cout, W): the ten convolutions of one decoder (encoders.UnetDecoder for efficientnet-b4: in (448,256,128,64,32) + skip (160,56,32,48,0))
SHAPES = [(608, 256, 4), (256, 256, 4), (312, 128, 8), (128, 128, 8), (160, 64, 16), (64, 64, 16), (112, 32, 32), (32, 32, 32), (32, 16, 64),
          (16, 16, 64)]


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _run(B, Cin, Cout, W, seed=0):
    from srbh_amd import encoders as E
    from srbh_amd import hrfuse as H
    g = torch.Generator().manual_seed(seed + Cin + W)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1, bias=False).to(DEV)
    x = torch.randn((B, Cin, W, W), generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn((B, Cout, W, W), generator=g).to(DEV)
    with H.head_precision("f16"):
        y = E.decoder_conv(conv, x)
        assert y.grad_fn is not None and "DecoderConv" in type(y.grad_fn).__name__      # the libsrbh path ran
        y.backward(gy)
    return conv, x, gy, y


@pytest.mark.parametrize("Cin,Cout,W", SHAPES)
@pytest.mark.parametrize("B", [3, 20])
def test_decoder_conv_forward_and_gradients(Cin, Cout, W, B):
    if B == 20 and W >= 32:
        B = 6
    conv, x, gy, y = _run(B, Cin, Cout, W)
    w = conv.weight.detach()
    xd = x.detach()
    h = lambda t: t.half().double()                 # noqa: E731
    b = lambda t: t.bfloat16().double()             # noqa: E731
    # forward: fp16 operands, fp32 accumulate
    ref = F.conv2d(h(xd), h(w), None, 1, 1)
    assert rel(y, ref) <= 2e-6
    assert rel(y, F.conv2d(xd.double(), w.double(), None, 1, 1)) <= 2e-3
    # data gradient: bf16 operands
    xr = xd.double().requires_grad_(True)
    F.conv2d(xr, b(w), None, 1, 1).backward(b(gy))
    assert rel(x.grad, xr.grad) <= 2e-6
    # weight gradient: bf16 operands
    wr = w.double().requires_grad_(True)
    F.conv2d(b(xd), wr, None, 1, 1).backward(b(gy))
    assert rel(conv.weight.grad, wr.grad) <= 5e-6
    # ... and against the unrounded graph at the bf16 level
    xr2, wr2 = xd.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(xr2, wr2, None, 1, 1).backward(gy.double())
    assert rel(x.grad, xr2.grad) <= 8e-3 and rel(conv.weight.grad, wr2.grad) <= 8e-3


@pytest.mark.parametrize("B,Cin,Cout,W", [(1, 8, 16, 4), (17, 24, 40, 4), (5, 40, 24, 8), (2, 8, 8, 16), (1, 16, 48, 64), (33, 16, 16, 4)])
def test_odd_channel_counts_and_ragged_batches(B, Cin, Cout, W):
    """channel counts that are not multiples of 16 (zero-padded while staged / masked on store), batches that do not fill the last
    tile of whole images"""
    conv, x, gy, y = _run(B, Cin, Cout, W, seed=3)
    w, xd = conv.weight.detach(), x.detach()
    assert rel(y, F.conv2d(xd.half().double(), w.half().double(), None, 1, 1)) <= 2e-6
    xr = xd.double().requires_grad_(True)
    F.conv2d(xr, w.bfloat16().double(), None, 1, 1).backward(gy.bfloat16().double())
    assert rel(x.grad, xr.grad) <= 2e-6
    wr = w.double().requires_grad_(True)
    F.conv2d(xd.bfloat16().double(), wr, None, 1, 1).backward(gy.bfloat16().double())
    assert rel(conv.weight.grad, wr.grad) <= 5e-6


def test_policy_and_fallbacks():
    """the exact-fp32 modes and unsupported planes keep the stock convolution and are COUNTED (encoders.STOCK_OPS); weight updates behind
    the version counter (fused optimizer: wcache.stamp) reach the packed images"""
    from srbh_amd import encoders as E
    from srbh_amd import hrfuse as H
    from srbh_amd import wcache
    conv = torch.nn.Conv2d(16, 16, 3, padding=1, bias=False).to(DEV)
    x = torch.randn(2, 16, 8, 8, device=DEV)
    E.stock_ops_reset()
    with H.head_precision("f32"):
        y0 = E.decoder_conv(conv, x.clone().requires_grad_(True))
    assert "DecoderConv" not in type(y0.grad_fn).__name__ and E.stock_ops_summary()["by_site"] == {"decoder_conv3x3": 1}
    with H.head_precision("f16"):
        y1 = E.decoder_conv(conv, x)
        assert rel(y1, y0) <= 2e-3
        y2 = E.decoder_conv(conv, torch.randn(2, 16, 12, 12, device=DEV))           # 12 x 12: not a power of two -> stock
        assert y2.shape == (2, 16, 12, 12) and E.stock_ops_summary()["calls"] == 2
        with torch.no_grad():
            conv.weight.mul_(2.0)            # bumps _version
            assert rel(E.decoder_conv(conv, x), 2 * y1) <= 1e-6
            torch._foreach_mul_([conv.weight], 0.5)      # (still bumps; the stamp path:)
            conv.weight.data.mul_(4.0)       # (a power of two: exact in fp16)
            wcache.stamp([conv.weight])      # what the optimizer hook / TrainStep's graph replay do for updates behind the version counter
            assert rel(E.decoder_conv(conv, x), 4 * y1) <= 1e-6
            key = conv._srbh_dconv_packs.kf
            E.decoder_conv(conv, x)
            assert conv._srbh_dconv_packs.kf == key          # unchanged weight: the pack is reused


def test_pack_table_equals_per_conv_packs():
    """DecoderPackTable: both images of all ten convs of a decoder in one launch == the per-conv srbh_hpack_conv_h16 packs, refreshed
    only when a weight changed; a decoder forward in the 16-bit mode uses it and matches the stock decoder at the fp16 level"""
    from srbh_amd import encoders as E
    from srbh_amd import hrfuse as H
    from srbh_amd import wcache
    torch.manual_seed(3)
    dec = E.UnetDecoder((8, 48, 32, 56, 160, 448), (256, 128, 64, 32, 16)).to(DEV).eval()
    feats = [torch.randn(2, c, 64 // s, 64 // s, device=DEV) for c, s in zip((8, 48, 32, 56, 160, 448), (1, 2, 4, 8, 16, 32))]
    with torch.no_grad():
        with H.head_precision("f32"):
            want = dec(*feats)
        with H.head_precision("f16"):
            got = dec(*feats)
            pt = dec._srbh_dpt
            assert len(pt.convs) == 10 and pt.key is not None
            for c in pt.convs:
                ref = E._DecoderConvPacks()
                pk = c._srbh_dconv_packs
                assert torch.equal(pk.f, ref.fwd(c.weight)) and torch.equal(pk.b, ref.bwd(c.weight))
            k0 = pt.key
            dec(*feats)
            assert pt.key is k0                      # nothing changed: no new pack launch
            pt.convs[3].weight.data.mul_(2.0)
            wcache.stamp([pt.convs[3].weight])
            dec(*feats)
            assert pt.key != k0 and torch.equal(pt.convs[3]._srbh_dconv_packs.f, E._DecoderConvPacks().fwd(pt.convs[3].weight))
    assert rel(got, want) <= 3e-3


@pytest.mark.parametrize("B", [1, 5, 32])
def test_inference_batchnorm_relu_in_the_conv_store(B, monkeypatch):
    """Round 4: in eval mode a decoder block's BatchNorm (folded) + ReLU ride in the conv's store (srbh_dconv_fwd_epi) instead of a pass of
    their own: the whole U-Net decoder against the conv + separate affine pass (same fma, same bits expected; bound 1e-6), all ten conv
    shapes of the model, non-trivial running statistics; the call count says every block took the fused form."""
    from srbh_amd import encoders as E
    from srbh_amd import hrfuse as H
    torch.manual_seed(3)
    enc_ch = (8, 48, 32, 56, 160, 448)
    dec = E.UnetDecoder(enc_ch, (256, 128, 64, 32, 16), n_blocks=5, use_batchnorm=True, center=False, attention_type=None).to(DEV)
    feats = [torch.randn((B, c, 64 >> i, 64 >> i), device=DEV) for i, c in enumerate(enc_ch)]
    dec.train()
    with torch.no_grad(), H.head_precision("f16"):
        for _ in range(2):
            dec(*[f[:max(2, B)] if B >= 2 else torch.cat([f, f]) for f in feats])
    dec.eval()
    calls = {"n": 0}
    real = E._decoder_conv_bn_relu_eval

    def counting(conv, bn, x):
        y = real(conv, bn, x)
        calls["n"] += y is not None
        return y
    with torch.no_grad():
        assert H.head_h16()
        monkeypatch.setattr(E, "_decoder_conv_bn_relu_eval", counting)
        a = dec(*feats)
        assert calls["n"] == 10, calls
        monkeypatch.setattr(E, "DCONV_EVAL_EPI", False)
        b = dec(*feats)
        assert calls["n"] == 10
    assert a.shape == (B, 16, 64, 64) and rel(a, b) <= 1e-6 and float(a.min()) >= 0.0
