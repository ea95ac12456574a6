"""Training-mode BatchNorm + activation and squeeze-and-excitation kernels (csrc/srbh_mbconv.hip) through the C ABI (autograd Functions
in mbconv_autograd.py) against the stock fp32 ops of the same computation -- what efficientnet_pytorch's MBConvBlock.forward and smp's
Conv2dReLU do, as the reference runs them at mymodels.py:276-287.  Tolerance 2e-5 relative against the stock ops evaluated in fp64
(fp32 kernels, different summation order), and the stock fp32 ops themselves must be no closer than ~the same order."""
import copy

import pytest, torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-5


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _bn(C, g, dtype=torch.float32):
    bn = nn.BatchNorm2d(C, momentum=0.01, eps=1e-3)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    return bn.to(DEV).to(dtype).train()


def _stock(bn, x, act, res, drop):
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps)
    if act == "silu":
        y = F.silu(y)
    elif act == "relu":
        y = F.relu(y)
    if drop is not None:
        y = y * drop.view(-1, 1, 1, 1)
    return y if res is None else y + res


@pytest.mark.parametrize("B,C,H", [(6, 37, 2), (64, 20, 4), (5, 9, 8), (3, 5, 16), (2, 130, 2), (7, 3, 1), (64, 16, 8),
                                   (4, 6, 32), (3, 5, 64), (90, 2, 16), (33, 40, 32)])      # (last four: the large-plane path)
@pytest.mark.parametrize("act,with_res,with_drop", [("silu", False, False), ("relu", False, False), (None, True, True), (None, True, False),
                                                    (None, False, False)])
def test_bn_act_train_matches_stock_ops(B, C, H, act, with_res, with_drop):
    from srbh_amd import mbconv_autograd as MB
    g = torch.Generator().manual_seed(B * 1000 + C * 10 + H)
    bn = _bn(C, g)
    ref = copy.deepcopy(bn).double()
    x0 = (torch.randn((B, C, H, H), generator=g) * 1.5 + 0.4).to(DEV)
    res0 = torch.randn((B, C, H, H), generator=g).to(DEV) if with_res else None
    drop = (torch.floor(0.8 + torch.rand(B, generator=g)) / 0.8).to(DEV) if with_drop else None
    x = x0.clone().requires_grad_(True)
    res = res0.clone().requires_grad_(True) if with_res else None
    assert MB.supported(bn, x)
    y = MB.bn_act_train(bn, x, act, res, drop)
    xr = x0.double().requires_grad_(True)
    rr = res0.double().requires_grad_(True) if with_res else None
    yr = _stock(ref, xr, act, rr, drop.double() if with_drop else None)
    assert rel(y, yr) <= TOL
    assert rel(bn.running_mean, ref.running_mean) <= TOL and rel(bn.running_var, ref.running_var) <= TOL
    gy = torch.randn(y.shape, generator=g).to(DEV)
    y.backward(gy)
    yr.backward(gy.double())
    assert rel(x.grad, xr.grad) <= 5 * TOL               # (the input gradient subtracts two channel means: cancellation)
    assert rel(bn.weight.grad, ref.weight.grad) <= TOL
    assert rel(bn.bias.grad, ref.bias.grad) <= TOL
    if with_res:
        assert torch.equal(res.grad, gy)


@pytest.mark.parametrize("B,C,H", [(4, 6, 32), (3, 5, 64), (5, 9, 8)])
def test_bn_act_train_large_channel_offset(B, C, H):
    """round-3 ADVICE: channels with |mean| >> std (x = 100 + randn).  E[x^2] - mean^2 on raw fp32 sums loses ~(mean/std)^2 * 1e-7 of the
    variance; the large-plane kernel now sums (x - pivot), the small-plane kernel is two-pass: var / running_var / output match the
    float64 stock BatchNorm at the suite's tolerance."""
    from srbh_amd import mbconv_autograd as MB
    g = torch.Generator().manual_seed(77 + H)
    bn = _bn(C, g)
    ref = copy.deepcopy(bn).double()
    x = (100.0 + torch.randn((B, C, H, H), generator=g)).to(DEV).requires_grad_(True)
    assert MB.supported(bn, x)
    y = MB.bn_act_train(bn, x, None, None, None)
    yr = _stock(ref, x.detach().double(), None, None, None)
    assert rel(bn.running_var, ref.running_var) <= TOL and rel(bn.running_mean, ref.running_mean) <= TOL
    assert rel(y, yr) <= 20 * TOL          # (x itself carries 100 * 2^-24 = 6e-6 of rounding relative to its unit spread)


@pytest.mark.parametrize("B,C,SQ,H", [(6, 144, 6, 16), (64, 40, 10, 8), (5, 672, 28, 4), (3, 2688, 112, 2), (2, 19, 3, 2), (4, 24, 6, 32)])
def test_bn_swish_se_train_matches_stock_ops(B, C, SQ, H):
    from srbh_amd import mbconv_autograd as MB
    g = torch.Generator().manual_seed(B * 1000 + C + H)
    bn = _bn(C, g)
    red = nn.Conv2d(C, SQ, 1).to(DEV)
    exp = nn.Conv2d(SQ, C, 1).to(DEV)
    with torch.no_grad():
        red.weight.mul_(3.0)
        exp.weight.mul_(3.0)
    ref_bn, ref_red, ref_exp = copy.deepcopy(bn).double(), copy.deepcopy(red).double(), copy.deepcopy(exp).double()
    x0 = (torch.randn((B, C, H, H), generator=g) * 1.5 + 0.4).to(DEV)
    x = x0.clone().requires_grad_(True)
    assert MB.supported(bn, x) and MB.se_supported(red, exp)
    y = MB.bn_swish_se_train(bn, x, red, exp)
    xr = x0.double().requires_grad_(True)
    s = F.silu(F.batch_norm(xr, ref_bn.running_mean, ref_bn.running_var, ref_bn.weight, ref_bn.bias, True, ref_bn.momentum, ref_bn.eps))
    yr = torch.sigmoid(ref_exp(F.silu(ref_red(F.adaptive_avg_pool2d(s, 1))))) * s
    assert rel(y, yr) <= TOL
    assert rel(bn.running_var, ref_bn.running_var) <= TOL
    gy = torch.randn(y.shape, generator=g).to(DEV)
    y.backward(gy)
    yr.backward(gy.double())
    assert rel(x.grad, xr.grad) <= 5 * TOL
    for a, b in ((bn.weight, ref_bn.weight), (bn.bias, ref_bn.bias), (red.weight, ref_red.weight), (red.bias, ref_red.bias),
                 (exp.weight, ref_exp.weight), (exp.bias, ref_exp.bias)):
        assert rel(a.grad, b.grad) <= 2 * TOL


def test_unsupported_shapes_keep_the_stock_ops_and_eval_is_untouched():
    from srbh_amd import mbconv_autograd as MB
    g = torch.Generator().manual_seed(1)
    bn = _bn(8, g)
    assert MB.supported(bn, torch.zeros(2, 8, 32, 32, device=DEV))            # 32x32 planes: the two-launch path
    assert not MB.supported(bn, torch.zeros(2, 8, 9, 9, device=DEV))          # 81-element planes
    assert not MB.supported(bn, torch.zeros(2, 8, 6, 6, device=DEV))          # 36-element planes
    assert not MB.supported(bn.eval(), torch.zeros(2, 8, 4, 4, device=DEV))
    assert not MB.supported(bn.train(), torch.zeros(2, 8, 4, 4))              # CPU tensor
    with torch.no_grad():
        assert not MB.supported(bn, torch.zeros(2, 8, 4, 4, device=DEV))


def test_encoder_and_decoder_training_step_agrees_with_the_stock_path(monkeypatch):
    """whole EfficientNet-B4 encoder + U-Net decoder, training mode, kernels on vs off: outputs, every gradient, every running statistic"""
    from srbh_amd import encoders, mbconv_autograd as MB
    torch.manual_seed(11)
    enc = encoders.get_encoder("efficientnet-b4", in_channels=8, depth=5, weights=None).to(DEV).train()
    dec = encoders.UnetDecoder(enc.out_channels, (256, 128, 64, 32, 16), n_blocks=5).to(DEV).train()
    enc2, dec2 = copy.deepcopy(enc), copy.deepcopy(dec)
    x = torch.randn(4, 8, 64, 64, device=DEV)
    monkeypatch.setattr(torch, "rand", lambda *a, **k: torch.full(a[0], 0.5, device=k.get("device"), dtype=k.get("dtype")))
    calls = {"n": 0}
    real = MB._fwd

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    monkeypatch.setattr(MB, "_fwd", counting)
    real_mid = MB.mid_se_train

    def counting_mid(*a, **k):
        calls["mid"] = calls.get("mid", 0) + 1
        return real_mid(*a, **k)
    monkeypatch.setattr(MB, "mid_se_train", counting_mid)
    out = dec(*enc(x))
    out.square().mean().backward()
    # every BatchNorm of the encoder and the decoder: one launch each, or two of them inside one fused MBConv middle (round 4)
    assert calls["n"] + 2 * calls.get("mid", 0) >= 105 and calls.get("mid", 0) >= 20, calls
    monkeypatch.setattr(MB, "ENABLED", False)
    out2 = dec2(*enc2(x))
    out2.square().mean().backward()
    assert rel(out, out2) <= 2e-4
    big = max(float(p.grad.norm()) for p in enc2.parameters() if p.grad is not None)
    for (k, a), (_, b) in zip(list(enc.named_parameters()) + list(dec.named_parameters()),
                              list(enc2.named_parameters()) + list(dec2.named_parameters())):
        if b.grad is None:
            assert a.grad is None, k
            continue
        # 32 blocks of training-mode BatchNorm over 4 x (2x2 .. 16x16) samples amplify the 1e-6 differences between two fp32 evaluation
        # orders (the same bound test_gpu_model.py applies between the stock GPU and CPU ops); the tight per-kernel bounds are above
        if float(b.grad.norm()) > 1e-3 * big:    # (gradients that are ~0 by construction carry only noise)
            assert rel(a.grad, b.grad) <= 5e-2, (k, rel(a.grad, b.grad))
    for (k, a), (_, b) in zip(list(enc.named_buffers()) + list(dec.named_buffers()), list(enc2.named_buffers()) + list(dec2.named_buffers())):
        if a.dtype.is_floating_point:
            # (running statistics of the deep 2x2 stages: same amplification; a mean that is 0 by construction -- the BatchNorm of a
            #  convolution of a BatchNorm output -- is 1e-10 of pure rounding noise, hence the absolute term)
            assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-6, (k, rel(a, b))
        else:
            assert torch.equal(a, b), k


@pytest.mark.parametrize("B,Cx,Cs,H,W", [(3, 5, 7, 2, 2), (2, 16, 0, 32, 32), (4, 9, 3, 4, 6), (2, 448, 160, 2, 2)])
def test_up2_cat_matches_interpolate_and_cat_bit_exact_forward(B, Cx, Cs, H, W):
    from srbh_amd import mbconv_autograd as MB
    g = torch.Generator().manual_seed(B + Cx + H)
    x = torch.randn((B, Cx, H, W), generator=g).to(DEV).requires_grad_(True)
    skip = torch.randn((B, Cs, 2 * H, 2 * W), generator=g).to(DEV).requires_grad_(True) if Cs else None
    assert MB.up2_cat_supported(x, skip)
    y = MB.up2_cat(x, skip)
    xr = x.detach().clone().requires_grad_(True)
    sr = skip.detach().clone().requires_grad_(True) if Cs else None
    yr = F.interpolate(xr, scale_factor=2, mode="nearest")
    if Cs:
        yr = torch.cat([yr, sr], dim=1)
    assert torch.equal(y, yr)
    gy = torch.randn(yr.shape, generator=g).to(DEV)
    y.backward(gy)
    yr.backward(gy)
    assert rel(x.grad, xr.grad) <= 1e-6                  # (2x2 sums in a different order)
    if Cs:
        assert torch.equal(skip.grad, sr.grad)


@pytest.mark.parametrize("B,Cin,Cout,H", [(64, 272, 1632, 2), (5, 24, 144, 16), (3, 1632, 272, 2), (64, 32, 192, 16), (7, 56, 336, 8), (2, 960, 160, 4),
                                          (9, 19, 37, 4), (64, 2688, 448, 2), (3, 48, 24, 32), (256, 160, 960, 4), (256, 272, 1632, 2), (200, 336, 56, 8)])
def test_pointwise_conv_matches_stock_conv(B, Cin, Cout, H):
    """1x1 convolution kernels (csrc/srbh_pwconv.hip, fp32 MFMA) against F.conv2d evaluated in fp64: forward, input gradient, weight gradient"""
    from srbh_amd.encoders import _PointwiseConvFn
    g = torch.Generator().manual_seed(B + Cin + Cout)
    x0 = torch.randn((B, Cin, H, H), generator=g).to(DEV)
    w0 = (torch.randn((Cout, Cin, 1, 1), generator=g) / Cin ** 0.5).to(DEV)
    x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    y = _PointwiseConvFn.apply(x, w)
    xr, wr = x0.double().requires_grad_(True), w0.double().requires_grad_(True)
    yr = F.conv2d(xr, wr)
    assert y.shape == yr.shape and rel(y, yr) <= 2e-6
    with torch.no_grad():                                   # the forward that reads W^T (what the encoder uses)
        assert rel(_PointwiseConvFn.apply(x0, w0, w0.view(Cout, Cin).t().contiguous()), yr) <= 2e-6
    gy = torch.randn(yr.shape, generator=g).to(DEV)
    y.backward(gy)
    yr.backward(gy.double())
    assert rel(x.grad, xr.grad) <= 2e-6
    assert rel(w.grad, wr.grad) <= 2e-6


def test_transpose_many_and_encoder_refresh():
    """one launch transposes every 1x1 weight of the encoder; a changed weight is seen by the next forward (never a stale W^T)"""
    from srbh_amd import encoders
    torch.manual_seed(5)
    enc = encoders.get_encoder("efficientnet-b4", in_channels=8, depth=5, weights=None).to(DEV).train()
    x = torch.randn(2, 8, 64, 64, device=DEV)
    enc(x)
    pt = enc.__dict__["_srbh_pwt"]
    assert len(pt.convs) == 62
    for c in pt.convs:
        assert torch.equal(c.__dict__["_srbh_wt"], c.weight.detach().view(c.weight.shape[0], -1).t())
    c = pt.convs[7]
    with torch.no_grad():
        c.weight.mul_(2.0)                                  # version bump: the cached W^T is stale now
    xi = torch.randn(2, c.weight.shape[1], 4, 4, device=DEV)
    y = c(xi)                                               # direct call between refreshes: must not use the stale copy
    assert rel(y, F.conv2d(xi.double(), c.weight.detach().double())) <= 2e-6
    enc(x)
    assert torch.equal(c.__dict__["_srbh_wt"], c.weight.detach().view(c.weight.shape[0], -1).t())


@pytest.mark.parametrize("B,inp,H,K", [(64, 28, 4, 3), (7, 40, 4, 5), (64, 68, 2, 5), (5, 20, 8, 3), (33, 14, 8, 5), (3, 112, 2, 3)])
def test_fused_mbconv_middle_equals_the_separate_kernels(B, inp, H, K, monkeypatch):
    """srbh_mbconv_mid_fwd / _bwd (round 4): BatchNorm0 + SiLU -> depthwise -> BatchNorm1 + SiLU + pool of an MBConv block as ONE launch per
    direction, against the same block on the separate kernels (bn_act, depthwise, bn_act + pool: each already pinned against the stock
    ops above): block output, every parameter gradient, the input gradient and both BatchNorms' running statistics."""
    from srbh_amd import encoders as E
    from srbh_amd import mbconv_autograd as MB
    torch.manual_seed(B + inp + K)
    blk = E.MBConvBlock(inp, inp, K, 1, 6, 64).to(DEV).train()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(blk)
    x0 = torch.randn(B, inp, H, H, device=DEV)
    gy = torch.randn(B, inp, H, H, device=DEV)
    outs = []
    for fused, m in ((True, blk), (False, ref)):
        monkeypatch.setattr(MB, "MID_FUSED", fused)
        x = x0.clone().requires_grad_(True)
        y = m(x)
        names = {type(f).__name__ for f in _graph_fns(y.grad_fn)}
        assert ("_MidSEFnBackward" in names) == fused, names
        y.backward(gy)
        outs.append((y.detach(), x.grad, {k: p.grad for k, p in m.named_parameters()}, {k: b.clone() for k, b in m.named_buffers()}))
    (y1, gx1, g1, b1), (y0, gx0, g0, b0) = outs
    assert rel(y1, y0) <= 2e-6 and rel(gx1, gx0) <= 2e-5
    for k in g0:
        assert rel(g1[k], g0[k]) <= 5e-5, k
    for k in b0:
        if b0[k].dtype.is_floating_point:
            assert rel(b1[k], b0[k]) <= 1e-6, k


def _graph_fns(fn, seen=None):
    seen = set() if seen is None else seen
    if fn is None or fn in seen:
        return seen
    seen.add(fn)
    for nxt, _ in fn.next_functions:
        _graph_fns(nxt, seen)
    return seen


def test_skip_gradient_through_the_expand_conv_node():
    """an MBConv block's skip connection routed through its expand conv's autograd node (_pointwise_with_skip): the gradient arriving over the
    skip and the conv's input gradient meet inside srbh_pwconv_bwd_data_res instead of in an add launched by autograd -- same dX / dW as the
    plain graph (conv(x) * a + x * b), every kernel form of the input gradient"""
    import torch
    from srbh_amd import encoders as E
    dev = "cuda:0"
    g = torch.Generator().manual_seed(4)
    for B, ci, co, hw in [(8, 24, 144, 32), (8, 160, 960, 4), (8, 272, 1632, 2), (8, 448, 2688, 2), (4, 56, 336, 8)]:
        conv = E.SamePadConv2d(ci, co, 1, hw, bias=False).to(dev)
        x0 = torch.randn((B, ci, hw, hw), generator=g).to(dev)
        a = torch.randn((B, co, hw, hw), generator=g).to(dev)
        b = torch.randn((B, ci, hw, hw), generator=g).to(dev)
        grads = []
        for through in (False, True):
            x = x0.clone().requires_grad_(True)
            conv.weight.grad = None
            xin = x * 1.0
            if through:
                y, s = E._pointwise_with_skip(conv, xin)
                assert s is not None
            else:
                y, s = conv(xin), xin
            ((y * a).sum() + (s * b).sum()).backward()
            grads.append((x.grad.clone(), conv.weight.grad.clone()))
        assert torch.equal(grads[0][1], grads[1][1])                                    # dW: the same kernel on the same operands
        assert float((grads[0][0] - grads[1][0]).abs().max()) <= 2e-6 * float(grads[0][0].abs().max())   # dX: the add moved into the store
