"""GPU: the 16-bit-tensor forms of the head's training kernels (round 3: bf16 gradient tensors / fp16 saved activations inside a
BasicBlock's backward, include/srbh.h SRBH_IO_* / SRBH_WG_* / SRBH_BN_* / SRBH_BAR_*) against their fp32-tensor forms on inputs that
are exactly representable in the 16-bit type: identical results where only the storage differs, one bf16 rounding where the output is
16-bit.  (Written after a first version of the bf16 widening read (e0, e1, e0, e1) out of every quad -- a __builtin_bit_cast of a single
ext-vector element -- while the loss of a training run still went down.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, C, HH, WW = 2, 16, 64, 128


def _setup():
    from srbh_amd import hrfuse as H
    torch.manual_seed(0)
    nh = H.to_nhwc
    g = nh((torch.randn(B, C, HH, WW, device=DEV) * 1e-5).to(torch.bfloat16).float())
    c = nh(torch.randn(B, C, HH, WW, device=DEV).to(torch.float16).float())
    out = nh(torch.randn(B, C, HH, WW, device=DEV))
    v = lambda s, o: torch.rand(C, device=DEV) * s + o          # noqa: E731
    return H, g, c, out, dict(mean=v(0.2, -0.1), invstd=v(1, 0.5), gamma=v(1, 0.5), s1=v(1, 0.5), h1=v(0.2, -0.1))


def _as16(H, t, dt):
    o = H.empty_nhwc(*t.shape, t.device, dt)
    o.copy_(t)
    return o


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))


@pytest.fixture(autouse=True)
def _mixed_mode():
    from srbh_amd import hrfuse as H
    H.set_head_precision("f16")
    yield
    H.set_head_precision("auto")


def test_bn_backward_16bit_tensors():
    from srbh_amd import hrfuse_autograd as HA
    H, g, c, out, k = _setup()
    c16, g16 = _as16(H, c, torch.float16), _as16(H, g, torch.bfloat16)
    r32 = HA.bn_backward(g, c, k["mean"], k["invstd"], k["gamma"], None, True, relu_ref=out)
    r16 = HA.bn_backward(g, c16, k["mean"], k["invstd"], k["gamma"], None, True, relu_ref=out, out_b16=True)
    assert r16[0].dtype == torch.bfloat16 and r16[3].dtype == torch.bfloat16
    assert _rel(r16[3], r32[3]) == 0.0                                  # dz = masked g: bf16-representable, exact
    assert _rel(r16[0], r32[0]) <= 3e-3 and _rel(r16[1], r32[1]) <= 1e-6 and _rel(r16[2], r32[2]) <= 1e-6
    r32 = HA.bn_backward(g, c, k["mean"], k["invstd"], k["gamma"], (k["s1"], k["h1"]), True)
    r16 = HA.bn_backward(g16, c16, k["mean"], k["invstd"], k["gamma"], (k["s1"], k["h1"]), True, out_b16=True)
    r16f = HA.bn_backward(g16, c16, k["mean"], k["invstd"], k["gamma"], (k["s1"], k["h1"]), True)
    assert _rel(r16f[0], r32[0]) <= 1e-6 and r16f[0].dtype == torch.float32          # 16-bit in, fp32 out: the same numbers
    assert _rel(r16[0], r32[0]) <= 3e-3 and _rel(r16[1], r32[1]) <= 1e-6 and _rel(r16[2], r32[2]) <= 1e-6


@pytest.mark.parametrize("cin,ks", [(16, 3), (64, 3), (32, 3), (32, 1), (64, 1)])
def test_data_gradient_16bit_tensors(cin, ks):
    from srbh_amd import hrfuse_autograd as HA
    H, g, _, _, _ = _setup()
    cv = torch.nn.Conv2d(cin, 16, ks, 1, ks // 2, bias=False).to(DEV)
    res = H.to_nhwc((torch.randn(B, cin, HH, WW, device=DEV) * 1e-5).to(torch.bfloat16).float())
    g16, res16 = _as16(H, g, torch.bfloat16), _as16(H, res, torch.bfloat16)
    want = HA.conv_dgrad(g, cv.weight, HA._PackedGrad(), res=res)
    assert _rel(HA.conv_dgrad(g16, cv.weight, HA._PackedGrad(), res=res16), want) == 0.0
    got = HA.conv_dgrad(g16, cv.weight, HA._PackedGrad(), res=res16, out_b16=True)
    assert got.dtype == torch.bfloat16 and _rel(got, want) <= 3e-3


@pytest.mark.parametrize("cin,ks", [(16, 3), (64, 3), (32, 3), (32, 1), (64, 1)])
def test_weight_gradient_16bit_tensors(cin, ks):
    from srbh_amd import hrfuse_autograd as HA
    H, g, c, _, k = _setup()
    g16 = _as16(H, g, torch.bfloat16)
    if cin == 16:          # the 16 -> 16 3x3 form also takes the fp16 saved activation (+ the producer's BatchNorm + ReLU)
        pre = (k["s1"], k["h1"], True)
        want = HA.conv_wgrad([c], pre, g, 16, 3)
        assert _rel(HA.conv_wgrad([_as16(H, c, torch.float16)], pre, g16, 16, 3), want) == 0.0
        assert _rel(HA.conv_wgrad([c], pre, g16, 16, 3), want) == 0.0
        assert _rel(HA.conv_wgrad([_as16(H, c, torch.float16)], None, g, 16, 3), HA.conv_wgrad([c], None, g, 16, 3)) == 0.0
    else:
        xs = H.to_nhwc(torch.randn(B, cin, HH, WW, device=DEV))
        assert _rel(HA.conv_wgrad([xs], None, g16, 16, ks), HA.conv_wgrad([xs], None, g, 16, ks)) == 0.0


def test_bn_add_relu_fp16_tensors():
    H, _, c, out, k = _setup()
    c16 = _as16(H, c, torch.float16)
    assert _rel(H.bn_add_relu(c16, k["s1"], k["h1"], out), H.bn_add_relu(c, k["s1"], k["h1"], out)) == 0.0
    assert _rel(H.bn_add_relu(c16, k["s1"], k["h1"], c16, k["gamma"], k["mean"]), H.bn_add_relu(c, k["s1"], k["h1"], c, k["gamma"], k["mean"])) == 0.0


@pytest.mark.parametrize("out_b16,with_mask", [(True, True), (False, True), (True, False)])
def test_bn_backward_sums_from_the_data_gradient_epilogue(out_b16, with_mask):
    """conv_dgrad(..., bstat=...) (srbh_hconv_args.bstat_*: the persistent 16 -> 16 kernel accumulates sum(dz), sum(dz * xhat) of its OWN
    output in the epilogue) against the separate srbh_bn_bwd_reduce pass over the same conv's output: the BatchNorm backward that follows
    must give the same dgamma / dbeta / dc.  (fp32 output: the sums are over identical values, only the summation order differs; bf16
    output: the epilogue sums the unrounded values, the reduce pass the rounded ones.)"""
    from srbh_amd import hrfuse_autograd as HA
    H, g, c, _, k = _setup()
    cv = torch.nn.Conv2d(16, 16, 3, 1, 1, bias=False).to(DEV)
    g16 = _as16(H, g, torch.bfloat16)
    mask = (k["s1"], k["h1"]) if with_mask else None
    assert HA.bstat_fusable(g16, cv.weight, c)
    da = HA.conv_dgrad(g16, cv.weight, HA._PackedGrad(), out_b16=out_b16)
    want = HA.bn_backward(da, c, k["mean"], k["invstd"], k["gamma"], mask, True, out_b16=out_b16)
    st = HA._stats_buf(16, c.device)
    da2 = HA.conv_dgrad(g16, cv.weight, HA._PackedGrad(), out_b16=out_b16,
                        bstat=(c, k["mean"], k["invstd"], None if mask is None else mask[0], None if mask is None else mask[1], st))
    assert torch.equal(da2, da)                                          # the conv's own output is untouched by the extra epilogue
    got = HA.bn_backward(da2, c, k["mean"], k["invstd"], k["gamma"], mask, True, out_b16=out_b16, stats_ready=st)
    tol = 2e-3 if out_b16 else 2e-5
    assert _rel(got[1], want[1]) <= tol and _rel(got[2], want[2]) <= tol      # dgamma, dbeta
    assert _rel(got[0], want[0]) <= (3e-3 if out_b16 else 2e-5)               # dc


@pytest.mark.parametrize("Bn,Cc,Hh,Ww", [(2, 16, 8, 12), (1, 4, 3, 5), (2, 6, 4, 4)])
def test_ps2_inverse_is_pixel_unshuffle(Bn, Cc, Hh, Ww):
    """srbh_ps2_inverse (the gradient's way back through nn.PixelShuffle(2), SR/HRfuse.py:23) == F.pixel_unshuffle, bit for bit: the
    16-elements-per-thread form (C % 4 == 0) and the one-element form (C = 6)"""
    from srbh_amd import hrfuse as H, hrfuse_autograd as HA
    g = torch.randn(Bn, Cc, 2 * Hh, 2 * Ww, device=DEV)
    got = HA.ps2_inverse(H.to_nhwc(g))
    assert torch.equal(got, torch.nn.functional.pixel_unshuffle(g, 2))
