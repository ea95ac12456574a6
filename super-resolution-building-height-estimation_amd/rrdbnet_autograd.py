"""Training path of RRDBNet: forward AND backward of the whole generator on libsrbh kernels (first slice).

The height stage never differentiates the RRDB stack (train.py:139-140,243-244); the SR stage does
(reference SR/rrdbnet_arch.py:538-592: ``l_g_total.backward()`` through ``self.net_g``).  This module is the autograd
``Function`` behind ``RRDBNet.forward`` / ``forward_feature`` when the training path is switched on
(``RRDBNet.enable_training_path()``; ``RealESRGAN(is_train=True)`` does it):

* forward  = the strict exact-fp32 network of ``RRDBNet._run_strict`` (v_mfma_f32_16x16x4_f32 convolutions with the
  LeakyReLU / ``x5*0.2+x`` / ``out*0.2+x`` epilogues fused), keeping every RDB's 192-channel dense buffer
  ``[x | x1 | x2 | x3 | x4]`` (post-activation, fp32 NHWC) -- 3 MiB per tile and RDB, i.e. 13 GB for the 23-block net at
  batch 64: nothing against 288 GB of HBM, so nothing is recomputed;
* backward = per conv: weight gradient as a GEMM over pixels (``srbh_hconv_wgrad_f32`` reading a strided view of the dense
  buffer), bias gradient, data gradient with the transposed + flipped weight through the same forward kernel, accumulated
  IN PLACE into the 192-channel gradient buffer by the conv's residual epilogue (out = conv + res1 on the same view); the
  LeakyReLU mask comes from the saved post-activation plane (x > 0 ? 1 : 0.2, as torch's leaky_relu backward).

Everything is exact fp32 (parity <= 1e-5 against the reference's autograd, fixture g14): the first slice trades speed
(~20x the fp16-MFMA inference path) for having the SR-stage generator step at all; an fp16/bf16-operand backward on the
trunk's MFMA kernel is the follow-up.  Element-wise glue (masks, 2x2 sums of the nearest-x2 backward, bias sums) uses torch
ops on the device.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, wcache
from . import hrfuse as H

__all__ = ["rrdbnet_apply", "set_train_precision"]

# Operand precision of the RRDBNet TRAINING graph: "f32" (default) = exact fp32 matrix cores in forward, data and weight gradients
# (what the <= 2e-3 / float64 parity tests pin); "mixed" = forward convs with fp16 operands, data and weight gradients with bf16
# operands, fp32 accumulation and fp32 residual / LeakyReLU epilogues everywhere (the head's TrainStep policy, hrfuse.py): the
# 16-bit matrix cores run these 64..192-channel convs several times faster.  SRBH_SR_TRAIN_PRECISION / set_train_precision().
import os as _os

_PRECISION = {"mode": _os.environ.get("SRBH_SR_TRAIN_PRECISION", "f32")}


def set_train_precision(mode):
    if mode not in ("f32", "mixed"):
        raise ValueError("RRDBNet training precision must be 'f32' or 'mixed'")
    _PRECISION["mode"] = mode


def _mixed():
    return _PRECISION["mode"] == "mixed"


class _Packs:
    """fp32 HWPACK32 images of one conv: forward pack (+ padded bias) and the transposed/flipped packs of its 64-channel
    input slices for the data gradient; rebuilt when the parameter changes."""

    def __init__(self):
        self.key = None

    def get(self, conv):
        w = conv.weight
        mixed = _mixed()
        key = (w._version, w.data_ptr(), conv.bias._version, conv.bias.data_ptr(), wcache.gen(w, conv.bias), mixed)
        if key != self.key:
            L = _lib.lib()
            cout, cin, ks, _ = w.shape
            wc = w.detach().float().contiguous()
            st = _lib.stream_ptr()
            self.mixed = mixed
            if mixed:          # fp16 forward pack, bf16 transposed / flipped packs for the data gradient (srbh_hpack_conv_h16)
                self.fwd = torch.empty(L.srbh_hpack_h16_bytes(cout, cin, ks) // 2, dtype=torch.float16, device=w.device)
                _lib.check(L.srbh_hpack_conv_h16(wc.data_ptr(), cout, cin, ks, 0, 0, self.fwd.data_ptr(), st), "hpack_h16(fwd)")
            else:
                self.fwd = torch.empty(L.srbh_hpack_bytes(cout, cin, ks) // 4, dtype=torch.float32, device=w.device)
                _lib.check(L.srbh_hpack_conv_f32(wc.data_ptr(), cout, cin, ks, 0, self.fwd.data_ptr(), st), "hpack(fwd)")
            self.bias = torch.zeros((cout + 15) // 16 * 16, dtype=torch.float32, device=w.device)
            self.bias[:cout] = conv.bias.detach().float()
            self.bwd = []          # [(c_lo, n, pack)]: data-gradient conv producing input channels c_lo .. c_lo + n
            for c_lo in range(0, cin, 64):
                n = min(64, cin - c_lo)
                sub = wc[:, c_lo:c_lo + n].contiguous()
                if mixed:
                    pk = torch.empty(L.srbh_hpack_h16_bytes(n, cout, ks) // 2, dtype=torch.float16, device=w.device)
                    _lib.check(L.srbh_hpack_conv_h16(sub.data_ptr(), n, cout, ks, 1, 1, pk.data_ptr(), st), "hpack_h16(bwd)")
                else:
                    pk = torch.empty(L.srbh_hpack_bytes(n, cout, ks) // 4, dtype=torch.float32, device=w.device)
                    _lib.check(L.srbh_hpack_conv_f32(sub.data_ptr(), n, cout, ks, 1, pk.data_ptr(), st), "hpack(bwd)")
                self.bwd.append((c_lo, n, pk))
            self.key = key
        wcache.keep(self.fwd, self.bias, self.bwd)
        return self


def _packs(conv) -> _Packs:
    return conv.__dict__.setdefault("_srbh_train_packs", _Packs()).get(conv)


def _conv(src, c0, ld0, pack, bias, cout, out, out_ld=0, out_coff=0, lrelu=False, res1=None, res2=None, bf16=False):
    """out[..., out_coff : out_coff + cout] = epilogue(conv3x3(src[..., :c0])): the exact-fp32 kernel, or -- `pack` is a 16-bit pack
    (mixed precision) -- the fp16 (forward) / bf16 (`bf16`: data gradients) operand form with the same fp32 epilogue."""
    a = _lib.HConvArgs()
    a.src0, a.c0, a.src0_ld = src.data_ptr(), c0, ld0
    a.w, a.bias = pack.data_ptr(), (bias.data_ptr() if bias is not None else None)
    a.cout, a.ksize = cout, 3
    a.B, a.H, a.W = src.shape[0], src.shape[1], src.shape[2]
    a.out, a.out_ld, a.out_coff, a.post_lrelu = out.data_ptr(), out_ld, out_coff, int(lrelu)
    if res1 is not None:
        a.res1, a.res1_ld, a.res1_scale = res1[0].data_ptr(), res1[1], res1[2]
    if res2 is not None:
        a.res2, a.res2_ld, a.res2_scale = res2[0].data_ptr(), res2[1], res2[2]
    if pack.dtype == torch.float16:
        _lib.check(_lib.lib().srbh_hconv_h16(C.byref(a), int(bf16), _lib.stream_ptr()), "hconv_h16(rrdbnet train)")
    else:
        _lib.check(_lib.lib().srbh_hconv_f32(C.byref(a), _lib.stream_ptr()), "hconv_f32(rrdbnet train)")


def _wgrad(src, c0, ld0, g, cout):
    """dW[cout][c0][3][3] = sum_px g[px][oc] * src[px + tap][ci]; src may be a strided view (ld0 floats per pixel)."""
    L = _lib.lib()
    dw = torch.empty((cout, c0, 3, 3), dtype=torch.float32, device=g.device)
    a = _lib.HWGradArgs()
    a.src0, a.c0, a.src0_ld = src.data_ptr(), c0, ld0
    a.dy, a.cout, a.ksize = g.data_ptr(), cout, 3
    a.B, a.H, a.W = g.shape[0], g.shape[1], g.shape[2]
    a.dw = dw.data_ptr()
    ws = torch.empty(L.srbh_hwgrad_ws_bytes(cout, c0, 3) // 4, dtype=torch.float32, device=g.device)
    a.ws = ws.data_ptr()
    if _mixed():           # (layers outside the bf16 kernel's granularity -- conv_first's 3 input channels -- are computed in fp32 inside)
        _lib.check(L.srbh_hconv_wgrad_b16(C.byref(a), _lib.stream_ptr()), "hconv_wgrad_b16(rrdbnet train)")
    else:
        _lib.check(L.srbh_hconv_wgrad_f32(C.byref(a), _lib.stream_ptr()), "hconv_wgrad_f32(rrdbnet train)")
    return dw


def _dgrad_into(g, cout, packs, dst, dst_ld, accumulate):
    """dst[..., :cin] (+)= conv^T(g, W), 64 input channels per launch; `accumulate`: add to what dst holds (residual epilogue)."""
    for c_lo, n, pk in packs.bwd:
        _conv(g, cout, cout, pk, None, n, dst, dst_ld, c_lo, res1=(dst[..., c_lo:], dst_ld, 1.0) if accumulate else None, bf16=True)


def _up2(t):
    Bn, Hn, Wn, Cn = t.shape
    o = torch.empty((Bn, 2 * Hn, 2 * Wn, Cn), dtype=torch.float32, device=t.device)
    _lib.check(_lib.lib().srbh_nearest2x_f32(t.data_ptr(), o.data_ptr(), Bn, 2 * Hn, 2 * Wn, Cn, _lib.stream_ptr()), "nearest2x")
    return o


def _up2_bwd(g):            # adjoint of nearest x2: the sum over each 2x2 block
    Bn, Hn, Wn, Cn = g.shape
    return g.reshape(Bn, Hn // 2, 2, Wn // 2, 2, Cn).sum(dim=(2, 4))


def _lrelu_mask_(g, y):     # g *= d lrelu(z)/dz with y = lrelu(z): y > 0 <=> z > 0 (torch: slope at z <= 0)
    return g.mul_(torch.where(y > 0, 1.0, 0.2))


class _RRDBNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, want_forward, *params):
        dev = x.device
        xs = x.detach().float().permute(0, 2, 3, 1).contiguous()          # (B,H,W,Cin) NHWC
        B, Hh, Ww, cin = xs.shape
        convs = {n: _packs(getattr(net, n)) for n in ("conv_first", "conv_body", "conv_up1", "conv_up2", "conv_hr", "conv_last")}
        feat = torch.empty((B, Hh, Ww, 64), dtype=torch.float32, device=dev)
        _conv(xs, cin, cin, convs["conv_first"].fwd, convs["conv_first"].bias, 64, feat)
        dense = []                                                          # one saved (B,H,W,192) buffer per RDB
        cur = feat
        for blk in net.body:
            x_rrdb = cur
            for r in (1, 2, 3):
                rdb = getattr(blk, f"rdb{r}")
                D = torch.empty((B, Hh, Ww, 192), dtype=torch.float32, device=dev)
                D[..., :64].copy_(cur)
                for k in range(1, 5):
                    p = _packs(getattr(rdb, f"conv{k}"))
                    c0 = 64 + 32 * (k - 1)
                    _conv(D, c0, 192, p.fwd, p.bias, 32, D, 192, c0, lrelu=True)
                p = _packs(rdb.conv5)
                nxt = torch.empty((B, Hh, Ww, 64), dtype=torch.float32, device=dev)
                _conv(D, 192, 192, p.fwd, p.bias, 64, nxt, res1=(D, 192, 0.2), res2=(x_rrdb, 64, 0.2) if r == 3 else None)
                dense.append(D)
                cur = nxt
        body = torch.empty_like(feat)
        _conv(cur, 64, 64, convs["conv_body"].fwd, convs["conv_body"].bias, 64, body, res1=(feat, 64, 1.0))
        t1 = _up2(body)
        u1 = torch.empty_like(t1)
        _conv(t1, 64, 64, convs["conv_up1"].fwd, convs["conv_up1"].bias, 64, u1, lrelu=True)
        t2 = _up2(u1)
        u2 = torch.empty_like(t2)
        _conv(t2, 64, 64, convs["conv_up2"].fwd, convs["conv_up2"].bias, 64, u2, lrelu=True)
        hr = torch.empty_like(u2)
        _conv(u2, 64, 64, convs["conv_hr"].fwd, convs["conv_hr"].bias, 64, hr, lrelu=bool(want_forward))
        ctx.net, ctx.want_forward, ctx.in_shape = net, bool(want_forward), tuple(x.shape)
        ctx.saved = (xs, feat, dense, cur, t1, u1, t2, u2, hr)
        if not want_forward:
            return hr.permute(0, 3, 1, 2)
        cout = net.conv_last.out_channels
        out = torch.empty((B, 4 * Hh, 4 * Ww, cout), dtype=torch.float32, device=dev)
        _conv(hr, 64, 64, convs["conv_last"].fwd, convs["conv_last"].bias, cout, out)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        xs, feat, dense, last, t1, u1, t2, u2, hr = ctx.saved
        grads = {}

        def conv_bwd(name_or_mod, src, c0, ld0, g, need_dx=True, dst=None, dst_ld=0, accumulate=False):
            """records dW / db of the conv; returns (or accumulates into dst) the gradient of its input"""
            mod = getattr(net, name_or_mod) if isinstance(name_or_mod, str) else name_or_mod
            p = _packs(mod)
            cout = mod.out_channels
            grads[id(mod.weight)] = _wgrad(src, c0, ld0, g, cout)
            grads[id(mod.bias)] = g.sum(dim=(0, 1, 2))
            if not need_dx:
                return None
            if dst is None:
                dst = torch.empty(src.shape[:3] + (c0,), dtype=torch.float32, device=g.device)
                dst_ld = c0
            _dgrad_into(g, cout, p, dst, dst_ld, accumulate)
            return dst

        g = gout.detach().float().permute(0, 2, 3, 1).contiguous()
        if ctx.want_forward:
            g = conv_bwd("conv_last", hr, 64, 64, g)
            _lrelu_mask_(g, hr)
        g = conv_bwd("conv_hr", u2, 64, 64, g)
        _lrelu_mask_(g, u2)
        g = _up2_bwd(conv_bwd("conv_up2", t2, 64, 64, g))
        _lrelu_mask_(g, u1)
        g_feat2 = _up2_bwd(conv_bwd("conv_up1", t1, 64, 64, g))           # gradient of feat + body_feat
        g = conv_bwd("conv_body", last, 64, 64, g_feat2)                  # ... flows into the body's output
        i = len(dense)
        for blk in reversed(list(net.body)):
            g_rrdb_out = g                                                # out = rdb3(.)*0.2 + x_rrdb
            g = g_rrdb_out * 0.2
            for r in (3, 2, 1):
                rdb = getattr(blk, f"rdb{r}")
                i -= 1
                D = dense[i]
                # x_next = conv5(D) * 0.2 + x : the conv sees 0.2 g, x (= D[..., :64]) sees g
                dD = torch.zeros_like(D)
                g5 = (g * 0.2).contiguous()
                conv_bwd(rdb.conv5, D, 192, 192, g5, dst=dD, dst_ld=192, accumulate=False)
                for k in (4, 3, 2, 1):
                    c0 = 64 + 32 * (k - 1)
                    gk = _lrelu_mask_(dD[..., c0:c0 + 32].contiguous(), D[..., c0:c0 + 32])
                    conv_bwd(getattr(rdb, f"conv{k}"), D, c0, 192, gk, dst=dD, dst_ld=192, accumulate=True)
                g = dD[..., :64] + g
            g = g + g_rrdb_out                                            # the RRDB's skip connection
        g = g + g_feat2                                                   # feat reaches the output through the trunk skip too
        need_dx = ctx.needs_input_grad[0]
        gx = conv_bwd("conv_first", xs, xs.shape[3], xs.shape[3], g.contiguous(), need_dx=need_dx)
        params = [p for p in net.parameters()]
        out = [gx.permute(0, 3, 1, 2).reshape(ctx.in_shape) if need_dx else None, None, None]
        for p in params:
            gp = grads.get(id(p))
            out.append(None if gp is None else gp.reshape(p.shape).to(p.dtype))
        return tuple(out)


def rrdbnet_apply(net, x, want_forward):
    """RRDBNet.forward / forward_feature with a recorded graph (x: (B,C,H,W) device tensor, any scale handled by the caller)."""
    params = list(net.parameters())
    return _RRDBNetFn.apply(x, net, bool(want_forward), *params)
