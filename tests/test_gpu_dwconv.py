"""Depthwise conv kernels (csrc/srbh_dwconv.hip) against the stock fp32 op of the same shape: forward, input gradient,
weight gradient, every (K, stride, static-same padding) combination the EfficientNet-B4 encoder uses, through the C ABI
(autograd Function in encoders.py).  Tolerance 1e-5 relative (fp32, different summation order)."""
import pytest, torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("K,stride,H,pad", [
    (3, 1, 32, (1, 1, 1, 1)), (3, 2, 32, (0, 1, 0, 1)), (5, 2, 16, (1, 2, 1, 2)), (5, 1, 8, (2, 2, 2, 2)), (3, 1, 4, (1, 1, 1, 1)),
    (5, 2, 4, (1, 2, 1, 2)), (5, 1, 2, (2, 2, 2, 2)), (3, 1, 2, (1, 1, 1, 1)), (5, 2, 7, (2, 2, 2, 2)), (3, 2, 9, (1, 1, 1, 1)),
])
def test_depthwise_conv_matches_stock_op(K, stride, H, pad):
    from srbh_amd.encoders import _DepthwiseConvFn
    g = torch.Generator().manual_seed(K * 100 + stride * 10 + H)
    B, C, W = 5, 37, H + (1 if H > 4 else 0)
    x = torch.randn((B, C, H, W), generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn((C, 1, K, K), generator=g) * 0.3).to(DEV).requires_grad_(True)
    y = _DepthwiseConvFn.apply(x, w, stride, pad)
    xr, wr = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, pad), wr, None, stride, 0, 1, C)
    assert y.shape == yr.shape
    assert rel(y, yr) <= 1e-5
    gy = torch.randn(yr.shape, generator=g).to(DEV)
    y.backward(gy)
    yr.backward(gy)
    assert rel(x.grad, xr.grad) <= 1e-5
    assert rel(w.grad, wr.grad) <= 1e-5


def test_encoder_uses_the_kernels_and_agrees_with_the_stock_path():
    from srbh_amd import encoders
    torch.manual_seed(3)
    enc = encoders.get_encoder("efficientnet-b4", in_channels=8, depth=5, weights=None).to(DEV).eval()
    dws = [m for m in enc.modules() if isinstance(m, encoders.SamePadConv2d) and m._depthwise]
    assert len(dws) == 32
    x = torch.rand((3, 8, 64, 64), device=DEV)
    with torch.no_grad():
        a = enc(x)
        for m in dws:
            m._depthwise = False
        b = enc(x)
    for u, v in zip(a, b):
        assert u.shape == v.shape and rel(u, v) <= 1e-5


def test_bad_arguments_are_errors():
    from srbh_amd import _lib
    x = torch.zeros((1, 2, 4, 4), device=DEV)
    w = torch.zeros((2, 1, 7, 7), device=DEV)
    y = torch.zeros((1, 2, 4, 4), device=DEV)
    rc = _lib.lib().srbh_dwconv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, 2, 4, 4, 7, 1, 3, 3, 4, 4, _lib.stream_ptr())
    assert rc != 0
    with pytest.raises(RuntimeError):
        _lib.check(rc, "dwconv_fwd")


def test_fused_inference_batchnorm_agrees_with_the_stock_path():
    """encoder + U-Net decoder in eval mode: BatchNorm(+SiLU/ReLU) as one libsrbh affine pass vs MIOpen's inference BN"""
    from srbh_amd import encoders
    from srbh_amd import hrfuse as H
    # (exact-fp32 decoder convs: with the round-4 fp16-operand decoder convs a 1e-7 difference between the two BatchNorm paths flips
    #  fp16 roundings downstream and shows as 3e-5 -- this test is about the BatchNorm kernels)
    with H.head_precision("f32"):
        _fused_bn_body()


def _fused_bn_body():
    from srbh_amd import encoders
    torch.manual_seed(5)
    enc = encoders.get_encoder("efficientnet-b4", in_channels=8, depth=5, weights=None).to(DEV)
    dec = encoders.UnetDecoder(enc.out_channels, (256, 128, 64, 32, 16), n_blocks=5, use_batchnorm=True, center=False,
                               attention_type=None).to(DEV)
    x = torch.rand((4, 8, 64, 64), device=DEV)
    enc.train(); dec.train()
    with torch.no_grad():
        for _ in range(2):
            dec(*enc(x))          # non-trivial running statistics
    enc.eval(); dec.eval()
    with torch.no_grad():
        a = dec(*enc(x))
        encoders.FUSED_BN_EVAL = False
        try:
            b = dec(*enc(x))
        finally:
            encoders.FUSED_BN_EVAL = True
    assert rel(a, b) <= 1e-5
    # a parameter update invalidates the cached affine
    with torch.no_grad():
        enc._bn0.weight.mul_(1.5)
        a2 = enc(x)[1]
        encoders.FUSED_BN_EVAL = False
        try:
            b2 = enc(x)[1]
        finally:
            encoders.FUSED_BN_EVAL = True
    assert rel(a2, b2) <= 1e-5

