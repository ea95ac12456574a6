"""GPU: srbh_hbwd16 -- the backward of a 3x3, 16 -> 16 conv behind its BatchNorm in ONE pass (csrc/srbh_hbwd16_kernel.h, round 5) -- against
(a) the three launches it replaces (srbh_bn_bwd_apply_io -> srbh_hconv_wgrad_b16 + srbh_hconv_h16(bf16)), same operand rounding: the
    bf16 data gradient equal up to rare one-ulp flips of the apply arithmetic, the fp32 results to 1e-5;
(b) a float64 evaluation of the same rounded operands;
and the whole BasicBlock backward (reference graph: SR/HRfuse.py:142-159 through torch autograd) with the fused passes on and off."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _nhwc(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


def _case(B, Hh, Ww, seed, mask):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
    gy = _nhwc(r(B, 16, Hh, Ww) * 1e-3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    c = _nhwc(r(B, 16, Hh, Ww) * 0.7 + 0.1)
    x = _nhwc(r(B, 16, Hh, Ww))
    mean, invstd = (r(16) * 0.1).to(DEV), (torch.rand(16, generator=g) + 0.5).to(DEV)
    coef, k1, k2 = (torch.rand(16, generator=g) + 0.5).to(DEV), (r(16) * 1e-5).to(DEV), (r(16) * 1e-5).to(DEV)
    ms = (torch.rand(16, generator=g) + 0.5).to(DEV) if mask else None
    mh = (r(16) * 0.2).to(DEV) if mask else None
    w = (r(16, 16, 3, 3) * 0.1).to(DEV)
    return gy, c, x, mean, invstd, (coef, k1, k2), (ms, mh) if mask else None, w


def _separate(gy, c, mean, invstd, consts, mask, x, pre, w, res, out_b16, bstat):
    """the three launches: apply (bf16 dc) -> weight gradient, data gradient"""
    import ctypes as C
    from srbh_amd import _lib
    from srbh_amd import hrfuse as H
    from srbh_amd import hrfuse_autograd as HA
    L = _lib.lib()
    B, Cc, Hh, Ww = c.shape
    dc = H.empty_nhwc(B, 16, Hh, Ww, c.device, torch.bfloat16)
    ms, mh = (mask[0].data_ptr(), mask[1].data_ptr()) if mask is not None else (None, None)
    _lib.check(L.srbh_bn_bwd_apply_io(gy.data_ptr(), c.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ms, mh, consts[0].data_ptr(), consts[1].data_ptr(),
                                      consts[2].data_ptr(), dc.data_ptr(), B * Hh * Ww, 16, 4 | 1, _lib.stream_ptr()), "apply")
    dw = HA.conv_wgrad([x], pre, dc, 16, 3)
    dx = HA.conv_dgrad(dc, w, HA._PackedGrad(), res=res, out_b16=out_b16, bstat=bstat)
    return dc, dx, dw


@pytest.mark.parametrize("B,Hh,Ww", [(2, 8, 64), (1, 12, 128), (3, 4, 192)])
def test_conv2_form_statistics_epilogue_bf16_out(B, Hh, Ww):
    from srbh_amd import hrfuse as H
    from srbh_amd import hrfuse_autograd as HA
    gy, c, x, mean, invstd, consts, _, w = _case(B, Hh, Ww, 7 + Hh, False)
    g = torch.Generator().manual_seed(99)
    s1, h1 = (torch.rand(16, generator=g) + 0.5).to(DEV), (torch.randn(16, generator=g) * 0.2).to(DEV)
    m1, i1 = (torch.randn(16, generator=g) * 0.1).to(DEV), (torch.rand(16, generator=g) + 0.5).to(DEV)
    with H.head_precision("f16"), torch.no_grad():
        assert HA.hbwd16_ok(c, x, w)
        st_a, st_b = HA._stats_buf(16, DEV), HA._stats_buf(16, DEV)
        dx, dw = HA.hbwd16(gy, c, mean, invstd, consts, None, x, (s1, h1, True), w, HA._PackedGrad(), out_b16=True, bstat=(x, m1, i1, s1, h1, st_a))
        dc, dx_r, dw_r = _separate(gy, c, mean, invstd, consts, None, x, (s1, h1, True), w, None, True, (x, m1, i1, s1, h1, st_b))
        torch.cuda.synchronize()
    assert dx.dtype == torch.bfloat16 and dx.shape == dx_r.shape
    assert _rel(dx.float(), dx_r.float()) <= 2e-3 and float((dx.float() != dx_r.float()).float().mean()) <= 0.02
    assert _rel(dw, dw_r) <= 1e-4
    assert _rel(st_a, st_b) <= 1e-4
    # float64 evaluation of the same rounded operands
    dcd = dc.double()
    xp = torch.relu(x.double() * s1.double().view(1, -1, 1, 1) + h1.double().view(1, -1, 1, 1)).to(torch.bfloat16).double()
    wr = w.to(torch.bfloat16).double()
    dx64 = F.conv_transpose2d(dcd, wr, padding=1)
    assert _rel(dx.float(), dx64) <= 6e-3                 # (one bf16 rounding of the output)
    dw64 = torch.nn.grad.conv2d_weight(xp, wr.shape, dcd, padding=1)
    assert _rel(dw, dw64) <= 1e-4


@pytest.mark.parametrize("B,Hh,Ww", [(2, 8, 64), (1, 4, 128)])
def test_conv1_form_mask_skip_gradient_fp32_out(B, Hh, Ww):
    from srbh_amd import hrfuse as H
    from srbh_amd import hrfuse_autograd as HA
    gy, c, x, mean, invstd, consts, mask, w = _case(B, Hh, Ww, 21 + Ww, True)
    res = _nhwc(torch.randn((B, 16, Hh, Ww), generator=torch.Generator().manual_seed(5)) * 1e-3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with H.head_precision("f16"), torch.no_grad():
        dx, dw = HA.hbwd16(gy, c, mean, invstd, consts, mask, x, None, w, HA._PackedGrad(), res=res, out_b16=False)
        dc, dx_r, dw_r = _separate(gy, c, mean, invstd, consts, mask, x, None, w, res, False, None)
        torch.cuda.synchronize()
    assert dx.dtype == torch.float32
    assert _rel(dx, dx_r) <= 1e-4 and _rel(dw, dw_r) <= 1e-4
    frac_masked = float((c * mask[0].view(1, -1, 1, 1) + mask[1].view(1, -1, 1, 1) <= 0).float().mean())
    assert 0.2 < frac_masked < 0.8
    dx64 = F.conv_transpose2d(dc.double(), w.to(torch.bfloat16).double(), padding=1) + res.double()
    assert _rel(dx, dx64) <= 1e-5
    dw64 = torch.nn.grad.conv2d_weight(x.to(torch.bfloat16).double(), w.shape, dc.double(), padding=1)
    assert _rel(dw, dw64) <= 1e-4


@pytest.mark.parametrize("inplanes", [16, 32])
def test_basicblock_backward_fused_equals_separate(inplanes, monkeypatch):
    """hrfuse.BasicBlock (SR/HRfuse.py:109-159) forward + backward in the mixed-precision training mode with srbh_hbwd16 on and off:
    plain 16-channel block = both fused passes, entry block (32 -> 16 with downsample) = conv2's only"""
    from srbh_amd import hrfuse as H
    from srbh_amd import hrfuse_autograd as HA
    from srbh_amd import _lib
    torch.manual_seed(3)
    blk = H.BasicBlock(inplanes, 16).to(DEV).train()
    assert (blk.downsample is None) == (inplanes == 16)
    x0 = torch.randn((2, inplanes, 8, 128), generator=torch.Generator().manual_seed(8)).to(DEV)
    gy = torch.randn((2, 16, 8, 128), generator=torch.Generator().manual_seed(9)).to(DEV) * 1e-3
    out = {}
    for flag in (False, True):
        monkeypatch.setattr(HA, "HBWD16", flag)
        for p_ in blk.parameters():
            p_.grad = None
        x = x0.clone().requires_grad_(True)
        with H.head_precision("f16"):
            y = HA.blocks_forward([blk], [x])
            y.backward(gy)
        torch.cuda.synchronize()
        out[flag] = [x.grad.clone()] + [p_.grad.clone() for p_ in blk.parameters()]
    for a, b in zip(out[True], out[False]):
        assert _rel(a, b) <= 5e-3, _rel(a, b)          # (bf16 gradient tensors: rare one-ulp flips of dc / da1 between the two paths)


def test_block_chain_handoff_equals_one_node_per_block(monkeypatch):
    """three BasicBlocks (32 -> 16 entry, two plain ones: HRfuse_residual.fuse, SR/HRfuse.py:181-183) as ONE autograd node with the masked
    bf16 hand-off between blocks (srbh_hbwd16 relu_bits form) against one node per block: the hand-off writes exactly the bits the
    reduce pass would have made of the fp32 gradient, so every gradient agrees to the noise of the atomically summed statistics"""
    from srbh_amd import hrfuse as H
    from srbh_amd import hrfuse_autograd as HA
    torch.manual_seed(11)
    blocks = [H.BasicBlock(32, 16).to(DEV).train(), H.BasicBlock(16, 16).to(DEV).train(), H.BasicBlock(16, 16).to(DEV).train()]
    gen = torch.Generator().manual_seed(4)
    xa, xb = torch.randn((2, 16, 8, 128), generator=gen).to(DEV), torch.randn((2, 16, 8, 128), generator=gen).to(DEV)
    gy = torch.randn((2, 16, 8, 128), generator=gen).to(DEV) * 1e-3
    out = {}
    for mode in ("nodes", "chain", "chain_nohandoff"):
        monkeypatch.setattr(HA, "BLOCK_CHAIN", mode != "nodes")
        monkeypatch.setattr(HA, "CHAIN_HANDOFF", mode == "chain")
        for b in blocks:
            for p_ in b.parameters():
                p_.grad = None
        a, b_ = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
        with H.head_precision("f16"):
            y = HA.blocks_forward(blocks, [a, b_])
            y.backward(gy)
        torch.cuda.synchronize()
        out[mode] = [y.detach().clone(), a.grad.clone(), b_.grad.clone()] + [p_.grad.clone() for b in blocks for p_ in b.parameters()]
    assert len(out["chain"]) == 3 + 9 + 6 + 6
    for mode in ("chain", "chain_nohandoff"):
        for u, v in zip(out[mode], out["nodes"]):
            assert _rel(u, v) <= 2e-3, (mode, _rel(u, v))
    assert torch.equal(out["chain"][0], out["nodes"][0])          # the forward is the same launches


def test_block_chain_releases_its_activations_after_backward():
    """the chain keeps its blocks' saved tensors in plain attributes of one autograd node (round-5 advice): they are dropped when its backward has
    run -- as autograd drops saved tensors -- and a second backward through the same graph says so instead of computing on freed state"""
    import pytest
    from srbh_amd import hrfuse as H
    from srbh_amd import hrfuse_autograd as HA
    torch.manual_seed(3)
    blocks = [H.BasicBlock(16, 16).to(DEV).train(), H.BasicBlock(16, 16).to(DEV).train()]
    x = torch.randn((2, 16, 8, 64), device=DEV, requires_grad=True)
    with H.head_precision("f16"):
        y = HA.blocks_forward(blocks, [x])
        node = y.grad_fn
        y.sum().backward(retain_graph=True)
        assert node.subs is None and x.grad is not None
        with pytest.raises(RuntimeError, match="second time"):
            y.sum().backward()


def test_deferred_weight_gradient_reduces_equal_the_immediate_ones():
    """srbh_hwgrad_defer / srbh_hwgrad_flush: several weight gradients (the generic bf16 kernel at 3x3 and 1x1, the fused entry pair,
    srbh_hbwd16) queue their ordered reduce and ONE pair of launches does them all: every dW bit-identical to the call that reduces at once
    (same kernels, same order of additions), workspaces kept alive until the flush."""
    from srbh_amd import hrfuse as H
    from srbh_amd import hrfuse_autograd as HA
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5)
    nh = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)      # noqa: E731
    B, Hh, Ww = 2, 64, 64
    x32 = nh(torch.randn((B, 32, Hh, Ww), generator=g))
    x16 = nh(torch.randn((B, 16, Hh, Ww), generator=g))
    c = nh(torch.randn((B, 16, Hh, Ww), generator=g))
    dy = nh(torch.randn((B, 16, Hh, Ww), generator=g) * 1e-2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    v = lambda s=1.0: (torch.rand(16, generator=g) * s + 0.5).to(dev)          # noqa: E731
    mean, invstd, consts = v(), v(), (v(), v(1e-5), v(1e-5))
    w = (torch.randn((16, 16, 3, 3), generator=g) * 0.1).to(dev)

    def all_grads():
        pg = HA._PackedGrad()
        out = [HA.conv_wgrad([x32], None, dy, 16, 3), HA.conv_wgrad([x32], None, dy, 16, 1)]
        out += list(HA.conv_wgrad_entry([x32], dy, dy, 16))
        out.append(HA.hbwd16(dy, c, mean, invstd, consts, None, x16, None, w, pg, out_b16=True)[1])
        out.append(HA.conv_wgrad([x16], None, dy, 16, 3))
        return out

    with H.head_precision("f16"), torch.no_grad():
        now = all_grads()
        torch.cuda.synchronize()
        with HA.deferred_wgrad_reduces():
            later = all_grads()
            # (allocations between the calls and the flush must not disturb the queued partial sums)
            junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(8)]
        torch.cuda.synchronize()
        del junk
    assert len(now) == len(later) == 6 and not HA._DEFER_KEEP and HA._DEFER_DEPTH[0] == 0
    for a, b in zip(now, later):
        assert a.shape == b.shape and bool(torch.isfinite(b).all()) and torch.equal(a, b)
