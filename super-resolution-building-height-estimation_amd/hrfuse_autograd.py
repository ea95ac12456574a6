"""Training path of the HR feature / fusion head: torch.autograd.Function wrappers whose forward AND backward
are libsrbh kernels (csrc/srbh_head.hip, csrc/srbh_head_bwd.hip).  torch.autograd only carries the graph.

The reference obtains these gradients from torch autograd over SR/HRfuse.py (train.py:254-256); parity is pinned by
tests/golden/g6_basicblock.npz and g7_head.npz (outputs, input grads, parameter grads, running statistics).
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib, wcache
from . import hrfuse as H


# ----------------------------------------------------------------------------- raw op wrappers
class _PackedGrad:
    """HWPACK32 of the data-gradient conv (transposed + flipped weight), cached per weight version."""

    def __init__(self):
        self.key = None
        self.w = None

    def get(self, weight, h16=False, gen_src=None):
        # (gen_src: `weight` is a fresh view of a parameter -- the optimizer's stamp lives on the parameter object, wcache.py)
        key = (weight._version, weight.data_ptr(), bool(h16), wcache.gen(weight if gen_src is None else gen_src))
        if key != self.key:
            L = _lib.lib()
            cout, cin, ks, _ = weight.shape
            wc = weight.detach().float().contiguous()
            if h16:
                buf = torch.empty(L.srbh_hpack_h16_bytes(cin, cout, ks) // 2, dtype=torch.float16, device=weight.device)
                _lib.check(L.srbh_hpack_conv_h16(wc.data_ptr(), cin, cout, ks, 1, 1, buf.data_ptr(), _lib.stream_ptr()), "hpack_conv_h16(T, bf16)")
            else:
                buf = torch.empty(L.srbh_hpack_bytes(cin, cout, ks) // 4, dtype=torch.float32, device=weight.device)
                _lib.check(L.srbh_hpack_conv_f32(wc.data_ptr(), cin, cout, ks, 1, buf.data_ptr(), _lib.stream_ptr()),
                           "hpack_conv_f32(T)")
            # (no host sync: `wc` is recycled by torch's stream-ordered allocator, and the pack kernel runs on that stream)
            self.key, self.w = key, buf
            if h16 and gen_src is None and isinstance(weight, nn.Parameter):     # refreshed behind the optimizer's step from now on (hrfuse.PACKS)
                import weakref
                wref = weakref.ref(weight)

                def rekey(wref=wref):
                    ww = wref()
                    return None if ww is None else (ww._version, ww.data_ptr(), True, wcache.gen(ww))
                H.PACKS.register(self, weight, buf, (cin, cout, ks, 1, 1), rekey)
        wcache.keep(self.w)
        return self.w


def _hconv_raw(srcs, w_packed, bias, cout, ks, pre=None, ps2=False, res=None, h16=False, out_b16=False, res_ld=0, bstat=None):
    """data-gradient conv (bf16 operands when h16): `srcs[0]` / `res` may be bf16 tensors (internal gradient tensors of a block),
    out_b16 writes one.  bstat = (c, mean, invstd, ms, mh, stats): the BatchNorm-backward sums of the conv's OUTPUT in its epilogue
    (srbh_hconv_args.bstat_*)"""
    L = _lib.lib()
    x0 = srcs[0]
    B, c0, Hh, Ww = x0.shape
    a = _lib.HConvArgs()
    a.src0, a.c0 = x0.data_ptr(), c0
    if pre is not None:
        a.pre_scale, a.pre_shift, a.pre_relu = pre[0].data_ptr(), pre[1].data_ptr(), int(pre[2])
    if len(srcs) > 1:
        a.src1, a.c1 = srcs[1].data_ptr(), srcs[1].shape[1]
    a.w = w_packed.data_ptr()
    a.bias = None if bias is None else bias.data_ptr()
    a.cout, a.ksize = cout, ks
    a.B, a.H, a.W = B, Hh, Ww
    a.pixelshuffle2 = int(ps2)
    out = (H.empty_nhwc(B, cout // 4, 2 * Hh, 2 * Ww, x0.device) if ps2
           else H.empty_nhwc(B, cout, Hh, Ww, x0.device, torch.bfloat16 if out_b16 else torch.float32))
    a.out = out.data_ptr()
    if res is not None:          # out = conv + res in the conv's epilogue (exact: fma(y, 1, res))
        a.res1, a.res1_ld, a.res1_scale = res.data_ptr(), res_ld or res.shape[1], 1.0     # (res_ld: a channel slice of a wider tensor)
    a.io_h16 = ((1 if x0.dtype == torch.bfloat16 else 0) | (4 if res is not None and res.dtype == torch.bfloat16 else 0)
                | (8 if out_b16 else 0))
    if a.io_h16 and not h16:
        raise RuntimeError("libsrbh data gradient: bf16 tensors need the bf16-operand mode")
    if bstat is not None:
        c, mean, invstd, ms, mh, st = bstat
        a.bstat_c, a.bstat_mean, a.bstat_invstd = c.data_ptr(), mean.data_ptr(), invstd.data_ptr()
        a.bstat_ms, a.bstat_mh = (None, None) if ms is None else (ms.data_ptr(), mh.data_ptr())
        a.stats, a.stats_clean = st.data_ptr(), int(H.stats_clean(st))
    if h16:      # (data gradients: bf16 operands -- fp32's exponent range, no loss scaling needed)
        _lib.check(L.srbh_hconv_h16(C.byref(a), 1, _lib.stream_ptr()), "hconv_h16(bf16)")
    else:
        _lib.check(L.srbh_hconv_f32(C.byref(a), _lib.stream_ptr()), "hconv_f32")
    return out


def bstat_fusable(g, weight, c):
    """can conv_dgrad(g, weight) also produce the BatchNorm-backward sums of its output w.r.t. the BatchNorm input `c`?  (the persistent
    16 -> 16 3x3 kernel in the 16-bit operand mode, fp32 `c`)"""
    cout, cin, ks, _ = weight.shape
    B, _, Hh, Ww = g.shape
    return (FUSE_BN_REDUCE and H.head_h16() and cout == 16 and cin == 16 and ks == 3 and Ww % 64 == 0 and Hh % 4 == 0 and c.dtype == torch.float32
            and H.bn_sync_world() <= 1)


def conv_dgrad(g, weight, cache: _PackedGrad, res=None, out_b16=False, res_ld=0, gen_src=None, bstat=None):
    """dX = conv^T(g, W) (+ res): the forward kernel with transposed + flipped weights; `res` (NHWC, same shape as dX) is the
    gradient arriving over a skip connection, added in the epilogue instead of by a separate pass.  16-bit operands (bf16:
    gradients need fp32's exponent range) only in the explicit "f16" head precision mode (H.set_head_precision)."""
    cout, cin, ks, _ = weight.shape
    h16 = H.head_h16()
    return _hconv_raw([g], cache.get(weight, h16, gen_src), None, cin, ks, res=res, h16=h16, out_b16=out_b16 and h16, res_ld=res_ld, bstat=bstat)


# ---- deferred weight-gradient reduces (srbh_hwgrad_defer / srbh_hwgrad_flush, include/srbh.h): inside `deferred_wgrad_reduces()` every
# weight-gradient call of this module queues the ordered reduce of its partial sums, and leaving the block runs all of them as ONE pair of
# launches.  The workspaces of queued jobs are kept alive here until then (torch's stream-ordered allocator would hand a freed one to the
# next kernel in front of the deferred reduce).  SRBH_WGRAD_DEFER=0: every call reduces at once (A/B aid).
WGRAD_DEFER = __import__("os").environ.get("SRBH_WGRAD_DEFER", "1") == "1"
_DEFER_KEEP = []
_DEFER_DEPTH = [0]


class deferred_wgrad_reduces:
    def __enter__(self):
        if WGRAD_DEFER:
            if _DEFER_DEPTH[0] == 0:
                _lib.check(_lib.lib().srbh_hwgrad_defer(1), "hwgrad_defer")
            _DEFER_DEPTH[0] += 1
        return self

    def __exit__(self, *exc):
        if WGRAD_DEFER:
            _DEFER_DEPTH[0] -= 1
            if _DEFER_DEPTH[0] == 0:
                try:
                    _lib.check(_lib.lib().srbh_hwgrad_flush(_lib.stream_ptr()), "hwgrad_flush")
                finally:
                    _DEFER_KEEP.clear()
        return False


def _keep_ws(*ts):
    if _DEFER_DEPTH[0]:
        _DEFER_KEEP.extend(ts)


def _wgrad_args(srcs, pre, g, cout, ks):
    L = _lib.lib()
    x0 = srcs[0]
    B, c0, Hh, Ww = x0.shape
    c1 = srcs[1].shape[1] if len(srcs) > 1 else 0
    dw = torch.empty((cout, c0 + c1, ks, ks), dtype=torch.float32, device=x0.device)
    a = _lib.HWGradArgs()
    a.src0, a.c0 = x0.data_ptr(), c0
    if pre is not None:
        a.pre_scale, a.pre_shift, a.pre_relu = pre[0].data_ptr(), pre[1].data_ptr(), int(pre[2])
    if c1:
        a.src1, a.c1 = srcs[1].data_ptr(), c1
    a.dy, a.cout, a.ksize = g.data_ptr(), cout, ks
    a.B, a.H, a.W = B, Hh, Ww
    a.dw = dw.data_ptr()
    ws = torch.empty(L.srbh_hwgrad_ws_bytes(cout, c0 + c1, ks) // 4, dtype=torch.float32, device=x0.device)
    a.ws = ws.data_ptr()
    a.io = (1 if x0.dtype == torch.float16 else 0) | (2 if g.dtype == torch.bfloat16 else 0)
    _keep_ws(ws, dw)
    return a, dw, ws


def conv_wgrad(srcs, pre, g, cout, ks):
    L = _lib.lib()
    a, dw, _ws = _wgrad_args(srcs, pre, g, cout, ks)
    if H.head_h16():       # mixed precision: bf16 operands (like the data gradients), fp32 accumulation
        _lib.check(L.srbh_hconv_wgrad_b16(C.byref(a), _lib.stream_ptr()), "hconv_wgrad_b16")
    else:
        _lib.check(L.srbh_hconv_wgrad_f32(C.byref(a), _lib.stream_ptr()), "hconv_wgrad_f32")
    return dw


def conv_wgrad_entry(srcs, g3, g1, cout):
    """(dW of the 3x3 conv1, dW of the 1x1 downsample conv) of a BasicBlock entry: both read the same input, so in the bf16-operand mode
    they are ONE pass over it (srbh_hconv_wgrad_entry_b16: the 1x1 gradient is one more product on the staged tile); exact-fp32 mode, or
    gradients of different element types: the two separate calls."""
    if not (H.head_h16() and g3.dtype == g1.dtype):
        return conv_wgrad(srcs, None, g3, cout, 3), conv_wgrad(srcs, None, g1, cout, 1)
    a3, dw3, _w3 = _wgrad_args(srcs, None, g3, cout, 3)
    a1, dw1, _w1 = _wgrad_args(srcs, None, g1, cout, 1)
    _lib.check(_lib.lib().srbh_hconv_wgrad_entry_b16(C.byref(a3), C.byref(a1), _lib.stream_ptr()), "hconv_wgrad_entry_b16")
    return dw3, dw1


FUSE_BN_REDUCE = __import__("os").environ.get("SRBH_FUSE_BN_REDUCE", "1") == "1"      # (0: the separate reduce pass, A/B aid)
# the backward of a 16 -> 16 conv behind its BatchNorm as ONE pass (srbh_hbwd16: dc never written); SRBH_HBWD16=0: apply + weight gradient +
# data gradient as three launches (A/B aid)
HBWD16 = __import__("os").environ.get("SRBH_HBWD16", "1") == "1"


def hbwd16_ok(c, x, weight):
    """shapes / element types srbh_hbwd16 takes (the gradient tensor is the bf16 one this backward writes itself): fp32 BatchNorm input and
    conv input, 16 -> 16 3x3, no synchronised statistics"""
    B, Cc, Hh, Ww = c.shape
    return (HBWD16 and H.head_h16() and c.dtype == torch.float32 and x.dtype == torch.float32 and tuple(weight.shape) == (16, 16, 3, 3)
            and Cc == 16 and tuple(x.shape) == tuple(c.shape) and H.bn_sync_world() <= 1 and bool(_lib.lib().srbh_hbwd16_supported(Hh, Ww)))


def hbwd16(g, c, mean, invstd, consts, mask, x, pre, weight, cache, res=None, out_b16=False, bstat=None, gen_src=None, relu_bits=None):
    """(dx, dw[, stats]) of conv behind BatchNorm in one pass: dc = coef*(g' - k1 - xhat*k2) formed while staged, dw = wgrad(x', dc),
    dx = conv^T(dc, W) (+ res).  consts = (coef, k1, k2) of bn_backward(apply=False); mask = (ms, mh) of the ReLU behind the BatchNorm or
    None; pre = (scale, shift, relu) of the conv's forward operand transform or None; bstat = (c', mean', invstd', ms', mh', stats):
    the BatchNorm-backward sums of dx (no res then).  relu_bits (with bstat = the PREVIOUS block's (c2', mean2', invstd2', None, None, stats) and
    out_b16): dx + res is masked with that block's closing ReLU, written as bf16 and summed for its bn2 -- its reduce pass, fused."""
    L = _lib.lib()
    B, Cc, Hh, Ww = c.shape
    dev = c.device
    a = _lib.HBwd16Args()
    a.g, a.c = g.data_ptr(), c.data_ptr()
    a.mean, a.invstd = mean.data_ptr(), invstd.data_ptr()
    a.coef, a.k1, a.k2 = consts[0].data_ptr(), consts[1].data_ptr(), consts[2].data_ptr()
    if mask is not None:
        a.mask_scale, a.mask_shift = mask[0].data_ptr(), mask[1].data_ptr()
    a.x = x.data_ptr()
    if pre is not None:
        a.pre_scale, a.pre_shift, a.pre_relu = pre[0].data_ptr(), pre[1].data_ptr(), int(pre[2])
    wp = cache.get(weight, True, gen_src)
    a.w = wp.data_ptr()
    a.B, a.H, a.W = B, Hh, Ww
    dx = H.empty_nhwc(B, 16, Hh, Ww, dev, torch.bfloat16 if out_b16 else torch.float32)
    a.dx, a.dx_b16 = dx.data_ptr(), int(out_b16)
    if res is not None:
        if res.dtype != torch.bfloat16:
            raise TypeError("srbh_hbwd16: the skip gradient must be a bf16 tensor")
        a.res = res.data_ptr()
    if bstat is not None:
        bc, bm, bi, bms, bmh, st = bstat
        a.bstat_c, a.bstat_mean, a.bstat_invstd = bc.data_ptr(), bm.data_ptr(), bi.data_ptr()
        if bms is not None:
            a.bstat_ms, a.bstat_mh = bms.data_ptr(), bmh.data_ptr()
        a.stats, a.stats_clean = st.data_ptr(), int(H.stats_clean(st))
    if relu_bits is not None:       # dx (+ res) continues through the previous block's closing ReLU and bn2: masked bf16 + that BatchNorm's sums
        a.relu_bits = relu_bits.data_ptr()
    dw = torch.empty((16, 16, 3, 3), dtype=torch.float32, device=dev)
    ws = torch.empty(L.srbh_hwgrad_ws_bytes(16, 16, 3) // 4, dtype=torch.float32, device=dev)
    a.dw, a.ws = dw.data_ptr(), ws.data_ptr()
    _keep_ws(ws, dw)
    _lib.check(L.srbh_hbwd16(C.byref(a), _lib.stream_ptr()), "hbwd16")
    return dx, dw


def _stats_buf(Cc, dev):
    return H.stats_acquire(Cc, dev)          # (zeroed pool: hrfuse.stats_acquire; released by _bwd_finalize(last=True))


def _bwd_finalize(st, Cc, count, gamma, invstd, dgamma, dbeta, coef, k1, k2, what, last=True):
    """srbh_bn_bwd_finalize; the LAST read of a pooled buffer also zeroes it and hands it back to the pool"""
    L = _lib.lib()
    p = lambda t: None if t is None else t.data_ptr()      # noqa: E731
    fin = L.srbh_bn_bwd_finalize_clear if (last and H.stats_clean(st)) else L.srbh_bn_bwd_finalize
    _lib.check(fin(st.data_ptr(), Cc, float(count), p(gamma), p(invstd), p(dgamma), p(dbeta), p(coef), p(k1), p(k2), _lib.stream_ptr()), what)
    if last:
        H.stats_release(st)


def channel_sum(g):
    """sum over (B,H,W) per channel of an NHWC tensor (bias gradient)."""
    L = _lib.lib()
    B, Cc, Hh, Ww = g.shape
    st = _stats_buf(Cc, g.device)
    _lib.check(L.srbh_bn_bwd_reduce(g.data_ptr(), None, None, None, None, None, B * Hh * Ww, Cc, st.data_ptr(),
                                    _lib.stream_ptr()), "bn_bwd_reduce")
    out = torch.empty(Cc, dtype=torch.float32, device=g.device)
    _bwd_finalize(st, Cc, 1.0, None, None, None, out, None, None, None, "bn_bwd_finalize")
    return out


def bn_backward(g, c, mean, invstd, gamma, mask, training, relu_ref=None, out_b16=False, stats_ready=None, apply=True):
    """BatchNorm (+ optional ReLU mask [c*ms+mh > 0]) backward.  Returns (dc, dgamma, dbeta).
    stats_ready: the statistics buffer the producer of `g` filled in its epilogue (conv_dgrad(..., bstat=...)): the reduce pass is skipped.
    relu_ref: the gradient first passes the block-closing ReLU (dz = g where relu_ref > 0) inside the reduce pass; returns
    (dc, dgamma, dbeta, dz).  16-bit tensors (TRAIN_IO16): g may be bf16, c fp16; out_b16 writes dz / dc as bf16.
    apply=False: no apply pass -- the first element returned is (coef, k1, k2), the per-channel constants of dc = coef*(dy - k1 - xhat*k2)."""
    L = _lib.lib()
    B, Cc, Hh, Ww = c.shape
    n = B * Hh * Ww
    dev = c.device
    st = _stats_buf(Cc, dev) if stats_ready is None else stats_ready
    ms, mh = (mask[0].data_ptr(), mask[1].data_ptr()) if mask is not None else (None, None)
    dz = None
    vec = Cc % 4 == 0 and 256 % (Cc // 4) == 0
    io_c = 2 if c.dtype == torch.float16 else 0
    ref_bits = relu_ref is not None and relu_ref.dtype == torch.int64
    use_io = vec and (io_c or out_b16 or g.dtype == torch.bfloat16 or ref_bits)
    if ref_bits and not vec:
        raise NotImplementedError("libsrbh BatchNorm backward: the ReLU bit pattern needs C % 4 == 0 and 256 % (C/4) == 0")
    if (io_c or out_b16 or g.dtype == torch.bfloat16) and not vec:
        raise NotImplementedError("libsrbh BatchNorm backward: 16-bit tensors need C % 4 == 0 and 256 % (C/4) == 0")
    odt = torch.bfloat16 if out_b16 else torch.float32
    if stats_ready is not None:
        assert relu_ref is None
    elif use_io:
        if relu_ref is not None:
            dz = H.empty_nhwc(B, Cc, Hh, Ww, dev, odt)
        io = io_c | (4 if g.dtype == torch.bfloat16 else 0) | (1 if out_b16 else 0) | (16 if H.stats_clean(st) else 0)
        if relu_ref is not None and relu_ref.dtype == torch.int64:      # the ReLU pattern as bits (hrfuse.bn_add_relu(want_bits=True))
            io |= 8
        _lib.check(L.srbh_bn_bwd_reduce_io(g.data_ptr(), None if relu_ref is None else relu_ref.data_ptr(),
                                           None if dz is None else dz.data_ptr(), c.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                           ms, mh, n, Cc, st.data_ptr(), io, _lib.stream_ptr()), "bn_bwd_reduce_io")
        if dz is not None:
            g = dz
    # (the fused reduce+ReLU pass exists in the 16-byte form only: a thread owns one 4-channel group and 256 threads must hold
    # whole pixels -- csrc/srbh_head_bwd.hip bn_bwd_reduce_impl; other widths, e.g. super_mid=24, take the two-pass fallback)
    elif relu_ref is not None and vec and mask is None:
        dz = torch.empty_like(relu_ref)
        _lib.check(L.srbh_bn_bwd_reduce_relu(g.data_ptr(), relu_ref.data_ptr(), dz.data_ptr(), c.data_ptr(), mean.data_ptr(),
                                             invstd.data_ptr(), n, Cc, st.data_ptr(), _lib.stream_ptr()), "bn_bwd_reduce_relu")
        g = dz
    else:
        if relu_ref is not None:
            g = dz = relu_mask(g, relu_ref)
        _lib.check(L.srbh_bn_bwd_reduce(g.data_ptr(), c.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ms, mh, n, Cc,
                                        st.data_ptr(), _lib.stream_ptr()), "bn_bwd_reduce")
    dgamma = torch.empty(Cc, dtype=torch.float32, device=dev)
    dbeta = torch.empty(Cc, dtype=torch.float32, device=dev)
    coef = torch.empty(Cc, dtype=torch.float32, device=dev)
    k1 = torch.empty(Cc, dtype=torch.float32, device=dev)
    k2 = torch.empty(Cc, dtype=torch.float32, device=dev)
    sync = training and H.bn_sync_world() > 1
    _bwd_finalize(st, Cc, n, gamma, invstd, dgamma, dbeta, coef, k1, k2, "bn_bwd_finalize", last=not sync)
    if sync:
        # synchronised statistics: dgamma / dbeta above stay the LOCAL sums (they are averaged with the other gradients),
        # the mean terms of dx are those of the global batch
        H.bn_allreduce_(st)
        scratch = torch.empty(2 * Cc, dtype=torch.float32, device=dev)
        _bwd_finalize(st, Cc, n * H.bn_sync_world(), gamma, invstd, scratch[:Cc], scratch[Cc:], coef, k1, k2, "bn_bwd_finalize(sync)")
    if not training:       # frozen statistics: the mean terms vanish
        k1.zero_()
        k2.zero_()
    if not apply:          # the consumer forms dc itself (hbwd16: srbh_hbwd16): (coef, k1, k2) instead of the tensor
        consts = (coef, k1, k2)
        return (consts, dgamma, dbeta, dz) if relu_ref is not None else (consts, dgamma, dbeta)
    dc = H.empty_nhwc(B, Cc, Hh, Ww, dev, odt)
    if use_io:
        io = io_c | (4 if g.dtype == torch.bfloat16 else 0) | (1 if out_b16 else 0)
        _lib.check(L.srbh_bn_bwd_apply_io(g.data_ptr(), c.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ms, mh, coef.data_ptr(),
                                          k1.data_ptr(), k2.data_ptr(), dc.data_ptr(), n, Cc, io, _lib.stream_ptr()), "bn_bwd_apply_io")
    else:
        _lib.check(L.srbh_bn_bwd_apply(g.data_ptr(), c.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ms, mh, coef.data_ptr(),
                                       k1.data_ptr(), k2.data_ptr(), dc.data_ptr(), n, Cc, _lib.stream_ptr()), "bn_bwd_apply")
    if relu_ref is not None:
        return dc, dgamma, dbeta, dz
    return dc, dgamma, dbeta


def relu_mask(g, ref):
    out = torch.empty_like(ref)
    _lib.check(_lib.lib().srbh_relu_mask_mul(g.data_ptr(), ref.data_ptr(), out.data_ptr(), ref.numel(), _lib.stream_ptr()),
               "relu_mask_mul")
    return out


def add_(a, b):
    _lib.check(_lib.lib().srbh_add_inplace(a.data_ptr(), b.data_ptr(), a.numel(), _lib.stream_ptr()), "add_inplace")
    return a


def ps2_inverse(g_ps):
    B, Cc, H2, W2 = g_ps.shape
    out = H.empty_nhwc(B, 4 * Cc, H2 // 2, W2 // 2, g_ps.device)
    _lib.check(_lib.lib().srbh_ps2_inverse(g_ps.data_ptr(), out.data_ptr(), B, H2 // 2, W2 // 2, Cc, _lib.stream_ptr()),
               "ps2_inverse")
    return out


def _bn_eval_stats(bn):
    return bn.running_mean.detach(), torch.rsqrt(bn.running_var.detach() + bn.eps)


# ----------------------------------------------------------------------------- conv (+bias, +PixelShuffle)
class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, conv, packed, gcache, ps2):
        xn = H.to_nhwc(x.detach())
        out, _ = H.hconv([xn], conv, packed, ps2=ps2)
        ctx.save_for_backward(xn, weight)
        ctx.ps2, ctx.gcache, ctx.has_bias = ps2, gcache, bias is not None
        ctx.pack_dx = xn.data_ptr() != x.data_ptr()   # producer was a stock (NCHW) op: hand it a packed gradient
        return out

    @staticmethod
    def backward(ctx, g):
        xn, weight = ctx.saved_tensors
        g = H.to_nhwc(g)
        if ctx.ps2:
            g = ps2_inverse(g)
        cout, cin, ks, _ = weight.shape
        dx = conv_dgrad(g, weight, ctx.gcache) if ctx.needs_input_grad[0] else None
        if dx is not None and ctx.pack_dx:
            dx = dx.contiguous()
        dw = conv_wgrad([xn], None, g, cout, ks) if ctx.needs_input_grad[1] else None
        db = channel_sum(g) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None, None, None


def conv_forward(conv: nn.Conv2d, packed, x, ps2=False):
    cache = conv.__dict__.setdefault("_srbh_gcache", _PackedGrad())
    return _ConvFn.apply(x, conv.weight, conv.bias, conv, packed, cache, ps2)


def upsampler_forward(up, x):
    for i, mod in enumerate(up):
        if isinstance(mod, nn.Conv2d):
            x = conv_forward(mod, up._packs.setdefault(i, H._PackedConv()), x, ps2=True)
    return x


# ----------------------------------------------------------------------------- BasicBlock
class _BasicBlockFn(torch.autograd.Function):
    """out = relu(bn2(conv2(relu(bn1(conv1(x))))) + idt(x)), x = cat(x0, x1)  (SR/HRfuse.py:142-159)."""

    @staticmethod
    def forward(ctx, blk, x0, x1, w1, g1, b1, w2, g2, b2, wd, gd, bd):
        return _bb_forward(ctx, blk, x0, x1, w1, g1, b1, w2, g2, b2, wd, gd, bd)

    @staticmethod
    def backward(ctx, g):
        with deferred_wgrad_reduces():
            return _bb_backward(ctx, g)


class _Handoff:
    """what a plain block's conv1 backward hands to the block in front of it inside a chain instead of an fp32 gradient tensor: the gradient
    already through that block's closing ReLU (bf16) and the partial sums of its bn2 backward"""

    def __init__(self, dz, stats):
        self.dz, self.stats = dz, stats


def _fuse2_ok(ctx):
    """this block's backward takes conv2 behind bn2 as one srbh_hbwd16 pass (and can therefore start from a _Handoff)"""
    sv = ctx.saved_tensors
    nsrc = ctx.nsrc
    c1, c2, ref = sv[nsrc], sv[nsrc + 1], sv[nsrc + 2]
    w2 = sv[nsrc + 11]
    return (bool(getattr(ctx, "io16", False)) and H.head_h16() and ctx.tr and FUSE_BN_REDUCE and hbwd16_ok(c2, c1, w2)
            and ref.dtype == torch.int64)


class _BBBodies:
    """(namespace of the two bodies; _BasicBlockFn and _BlockChainFn call them with a ctx or a ctx-like object)"""

    @staticmethod
    def forward(ctx, blk, x0, x1, w1, g1, b1, w2, g2, b2, wd, gd, bd):
        blk._check()
        tr = blk.training
        srcs = [H.to_nhwc(x0.detach())] + ([H.to_nhwc(x1.detach())] if x1 is not None else [])
        B, _, Hh, Ww = srcs[0].shape
        n = B * Hh * Ww
        fuse_entry = blk.downsample is not None and H.head_h16()     # conv1 + the 1x1 downsample conv: one pass over the input
        # 16-bit block-internal tensors (H.TRAIN_IO16): the 16-channel 3x3 blocks of the head, mixed-precision mode only
        io16 = (H.TRAIN_IO16 and H.head_h16() and blk.conv1.out_channels == 16 and blk.conv2.out_channels == 16 and Ww % 64 == 0
                and Hh % 4 == 0 and all(t.shape[1] % 16 == 0 for t in srcs) and (blk.downsample is not None or srcs[0].shape[1] == 16))
        ctx.io16 = io16
        a1_16, a2_16 = io16 and "c1" in H.TRAIN_IO16_ACT, io16 and "c2" in H.TRAIN_IO16_ACT
        if fuse_entry:
            c1, st1, d_f, std_f = H.hconv_entry(srcs, blk.conv1, blk._p1, blk.downsample[0], blk._pd, want_stats=tr, out_h16=a1_16)
        else:
            c1, st1 = H.hconv(srcs, blk.conv1, blk._p1, want_stats=tr, out_h16=a1_16)
        s1, h1, m1, i1 = H.bn_scale_shift(blk.bn1, st1, n, tr)
        c2, st2 = H.hconv([c1], blk.conv2, blk._p2, pre=(s1, h1, True), want_stats=tr, out_h16=a2_16)
        s2, h2, m2, i2 = H.bn_scale_shift(blk.bn2, st2, n, tr)
        if not tr:
            (m1, i1), (m2, i2) = _bn_eval_stats(blk.bn1), _bn_eval_stats(blk.bn2)
        d = md = idd = None
        if blk.downsample is not None:
            d, std = (d_f, std_f) if fuse_entry else H.hconv(srcs, blk.downsample[0], blk._pd, want_stats=tr, out_h16=a1_16)
            sd, hd, md, idd = H.bn_scale_shift(blk.downsample[1], std, n, tr)
            if not tr:
                md, idd = _bn_eval_stats(blk.downsample[1])
        Cc2 = c2.shape[1]
        bits = H.RELU_BITS and Cc2 % 4 == 0 and 256 % (Cc2 // 4) == 0
        if blk.downsample is not None:
            r = H.bn_add_relu(c2, s2, h2, d, sd, hd, want_bits=bits)
        else:
            r = H.bn_add_relu(c2, s2, h2, srcs[0], want_bits=bits)
        out, ref = r if bits else (r, r)         # what the backward's ReLU mask is read from: 1 bit per element, or the fp32 output
        ctx.blk, ctx.tr, ctx.nsrc = blk, tr, len(srcs)
        ctx.save_for_backward(*srcs, c1, c2, ref, s1, h1, m1, i1, m2, i2, w1, g1, w2, g2,
                              *([d, md, idd, wd, gd] if d is not None else []))
        return out

    @staticmethod
    def backward(ctx, g, handoff=None, prev=None):
        """handoff: a _Handoff from the block behind this one in a chain (then g is None); prev: the ctx of the block in FRONT of this one in
        a chain -- if both blocks qualify, dx0 comes back as a _Handoff for it instead of an fp32 tensor"""
        blk, tr, nsrc = ctx.blk, ctx.tr, ctx.nsrc
        sv = ctx.saved_tensors
        srcs = list(sv[:nsrc])
        c1, c2, out, s1, h1, m1, i1, m2, i2, w1, g1, w2, g2 = sv[nsrc:nsrc + 13]
        has_ds = len(sv) > nsrc + 13
        if has_ds:
            d, md, idd, wd, gd = sv[nsrc + 13:]
        caches = blk.__dict__.setdefault("_srbh_gcaches", [_PackedGrad(), _PackedGrad(), _PackedGrad()])
        b16 = bool(getattr(ctx, "io16", False)) and H.head_h16()       # gradient tensors internal to this backward: bf16
        # through the final ReLU (folded into bn2's reduce pass) -> bn2 -> conv2
        fuse2 = b16 and tr and FUSE_BN_REDUCE and hbwd16_ok(c2, c1, w2)
        consts1 = None
        if handoff is not None:
            assert fuse2 and g is None
            dz = handoff.dz      # (already through this block's closing ReLU; bn2's sums came with it)
            k2c, dg2, db2 = bn_backward(dz, c2, m2, i2, g2, None, tr, out_b16=True, stats_ready=handoff.stats, apply=False)
        else:
            g = H.to_nhwc(g)
        if fuse2:
            # conv2's whole backward behind bn2 in ONE pass (srbh_hbwd16): dc2 is never written; bn1's backward sums come out of its epilogue
            if handoff is None:
                k2c, dg2, db2, dz = bn_backward(g, c2, m2, i2, g2, None, tr, relu_ref=out, out_b16=True, apply=False)
            st1 = _stats_buf(16, c1.device)
            da1, dw2 = hbwd16(dz, c2, m2, i2, k2c, None, c1, (s1, h1, True), w2, caches[1], out_b16=True, bstat=(c1, m1, i1, s1, h1, st1))
            plain = (not has_ds and nsrc == 1 and ctx.needs_input_grad[1] and hbwd16_ok(c1, srcs[0], w1))
            r1 = bn_backward(da1, c1, m1, i1, g1, (s1, h1), tr, out_b16=True, stats_ready=st1, apply=not plain)
            if plain:
                consts1, dg1, db1 = r1
            else:
                dc1, dg1, db1 = r1
        else:
            dc2, dg2, db2, dz = bn_backward(g, c2, m2, i2, g2, None, tr, relu_ref=out, out_b16=b16)
            dw2 = conv_wgrad([c1], (s1, h1, True), dc2, w2.shape[0], 3)
        # relu -> bn1 -> conv1   (mask: bn1(c1) > 0): bn1's backward sums come out of conv2's data-gradient epilogue when that is the
        # persistent 16 -> 16 kernel (one read of c1 there instead of a reduce pass over da1 and c1)
        if fuse2:
            pass
        elif bstat_fusable(dc2, w2, c1):
            st1 = _stats_buf(c1.shape[1], c1.device)
            da1 = conv_dgrad(dc2, w2, caches[1], out_b16=b16, bstat=(c1, m1, i1, s1, h1, st1))
            dc1, dg1, db1 = bn_backward(da1, c1, m1, i1, g1, (s1, h1), tr, out_b16=b16, stats_ready=st1)
        else:
            da1 = conv_dgrad(dc2, w2, caches[1], out_b16=b16)
            dc1, dg1, db1 = bn_backward(da1, c1, m1, i1, g1, (s1, h1), tr, out_b16=b16)
        need_dx = ctx.needs_input_grad[1] or (nsrc > 1 and ctx.needs_input_grad[2])
        dwd = dgd = dbd = None
        if consts1 is not None:
            # plain 16-channel block: conv1's backward behind bn1 in one pass as well: dc1 never written, the skip gradient dz added in the
            # epilogue, dx leaves as the fp32 tensor autograd carries
            if prev is not None and CHAIN_HANDOFF and _fuse2_ok(prev):
                # the block in front takes this gradient straight through its closing ReLU and bn2's reduce: masked bf16 + sums, no fp32 tensor
                psv, pn = prev.saved_tensors, prev.nsrc
                pc2, pref, pm2, pi2 = psv[pn + 1], psv[pn + 2], psv[pn + 7], psv[pn + 8]
                st2p = _stats_buf(16, c1.device)
                dzp, dw1 = hbwd16(da1, c1, m1, i1, consts1, (s1, h1), srcs[0], None, w1, caches[0], res=dz, out_b16=True,
                                  bstat=(pc2, pm2, pi2, None, None, st2p), relu_bits=pref)
                return (None, _Handoff(dzp, st2p), None, dw1, dg1, db1, dw2, dg2, db2, None, None, None)
            dx0, dw1 = hbwd16(da1, c1, m1, i1, consts1, (s1, h1), srcs[0], None, w1, caches[0], res=dz, out_b16=False)
            return (None, dx0, None, dw1, dg1, db1, dw2, dg2, db2, None, None, None)
        skip = dz if need_dx else None          # gradient arriving over the identity / downsample path
        if has_ds:
            dd, dgd, dbd = bn_backward(dz, d, md, idd, gd, None, tr, out_b16=b16)
            dw1, dwd = conv_wgrad_entry(srcs, dc1, dd, w1.shape[0]) if wd.shape[0] == w1.shape[0] else (
                conv_wgrad(srcs, None, dc1, w1.shape[0], 3), conv_wgrad(srcs, None, dd, wd.shape[0], 1))
            skip = conv_dgrad(dd, wd, caches[2], out_b16=b16) if need_dx else None
        else:
            dw1 = conv_wgrad(srcs, None, dc1, w1.shape[0], 3)
        dx0 = dx1 = None
        c0 = srcs[0].shape[1]
        if need_dx and nsrc == 2 and c0 == 16 and srcs[1].shape[1] == 16 and w1.shape[0] == 16 and H.head_h16():
            # two-source entry (cat(x_lr, x_hr), 16 + 16 channels): the data gradient as TWO 16 -> 16 convs over dc1 (the persistent
            # kernel: ~0.67 of the HBM peak, the 16 -> 32 template form ran at 0.28) that write dx_lr and dx_hr as two dense NHWC
            # tensors -- channel slices of one 32-channel tensor came back through autograd as strided views and cost an
            # nchw_to_nhwc pass each.  The skip gradient (32 channels) is read in place through its channel stride.
            halves = blk.__dict__.setdefault("_srbh_gcaches_split", [_PackedGrad(), _PackedGrad()])
            wa, wb = w1[:, :16], w1[:, 16:]
            ld = skip.shape[1] if skip is not None else 0
            dx0 = conv_dgrad(dc1, wa, halves[0], res=None if skip is None else skip[:, :16], res_ld=ld, gen_src=w1)
            dx1 = conv_dgrad(dc1, wb, halves[1], res=None if skip is None else skip[:, 16:], res_ld=ld, gen_src=w1)
        elif need_dx:
            dx = conv_dgrad(dc1, w1, caches[0], res=skip)
            if nsrc == 1:
                dx0 = dx
            else:
                dx0, dx1 = dx[:, :c0], dx[:, c0:]
        return (None, dx0, dx1, dw1, dg1, db1, dw2, dg2, db2, dwd, dgd, dbd)


_bb_forward, _bb_backward = _BBBodies.forward, _BBBodies.backward
# a Sequential of BasicBlocks as ONE autograd node (round 5): the gradient between two blocks never exists as an fp32 tensor when both take
# the fused passes -- the conv1 backward of the later block writes it masked, as bf16, together with the earlier block's bn2 sums
# (srbh_hbwd16 relu_bits form).  SRBH_BLOCK_CHAIN=0: one node per block (A/B aid; same arithmetic up to bf16 rounding of that gradient)
BLOCK_CHAIN = __import__("os").environ.get("SRBH_BLOCK_CHAIN", "1") == "1"
CHAIN_HANDOFF = __import__("os").environ.get("SRBH_CHAIN_HANDOFF", "1") == "1"


class _SubCtx:
    """what the block bodies need of an autograd ctx"""

    def __init__(self, needs):
        self.needs_input_grad = needs
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def _block_params(blk):
    ds = blk.downsample
    return [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias] + (
        [] if ds is None else [ds[0].weight, ds[1].weight, ds[1].bias])


class _BlockChainFn(torch.autograd.Function):
    """blocks[-1](... blocks[0](cat(x0, x1))) (nn.Sequential of BasicBlocks: SR/HRfuse.py:164-169,181-183), the per-block bodies unchanged"""

    @staticmethod
    def forward(ctx, blocks, x0, x1, *params):
        subs, off, cur0, cur1 = [], 0, x0, x1
        for k, blk in enumerate(blocks):
            n = 6 if blk.downsample is None else 9
            pr = list(params[off:off + n]) + [None] * (9 - n)
            off += n
            need0 = True if k > 0 else bool(ctx.needs_input_grad[1])
            need1 = bool(k == 0 and x1 is not None and ctx.needs_input_grad[2])
            sub = _SubCtx((False, need0, need1) + (True,) * 9)
            cur0, cur1 = _bb_forward(sub, blk, cur0, cur1, *pr), None
            subs.append(sub)
        ctx.subs, ctx.counts = subs, [6 if b.downsample is None else 9 for b in blocks]
        return cur0

    @staticmethod
    def backward(ctx, g):
        subs = ctx.subs
        if subs is None:
            raise RuntimeError("srbh BasicBlock chain: backward a second time through the same graph -- the saved activations were released "
                               "after the first one (as autograd releases saved tensors; retain_graph=True is not supported here: run the "
                               "forward again, or SRBH_BLOCK_CHAIN=0 for one autograd node per block)")
        # (the blocks' saved tensors live in plain attributes, not in ctx.save_for_backward -- a chain of bodies shares one node --, so they are
        #  released HERE: otherwise ~1 GB of activations per head stayed referenced until the graph node itself died)
        ctx.subs = None
        pgrads = [None] * len(subs)
        handoff, dx0, dx1 = None, None, None
        with deferred_wgrad_reduces():           # the ~7 weight gradients of the chain reduce their partial sums in one pair of launches at the end
            for k in range(len(subs) - 1, -1, -1):
                r = _bb_backward(subs[k], None if handoff is not None else g, handoff=handoff, prev=subs[k - 1] if k > 0 else None)
                pgrads[k] = list(r[3:3 + ctx.counts[k]])
                if isinstance(r[1], _Handoff):
                    handoff, g = r[1], None
                else:
                    handoff, g = None, r[1]
                dx0, dx1 = r[1], r[2]
        flat = [t for pg in pgrads for t in pg]
        return (None, dx0, dx1, *flat)


def blocks_forward(blocks, inputs):
    x0 = inputs[0]
    x1 = inputs[1] if len(inputs) > 1 else None
    blocks = list(blocks)
    if BLOCK_CHAIN and len(blocks) > 1:
        return _BlockChainFn.apply(blocks, x0, x1, *[p for b in blocks for p in _block_params(b)])
    for blk in blocks:
        ds = blk.downsample
        x0 = _BasicBlockFn.apply(blk, x0, x1, blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight,
                                 blk.bn2.weight, blk.bn2.bias,
                                 None if ds is None else ds[0].weight, None if ds is None else ds[1].weight,
                                 None if ds is None else ds[1].bias)
        x1 = None
    return x0


# ---- head precision: everything inside these Functions belongs to a recorded graph (training).  torch switches grad mode
# off inside Function.forward / backward, so the "auto" precision rule (fp16 operands only for inference) is told explicitly.
def _exact(fn):
    def wrapped(*a, **k):
        H._HEAD_PRECISION["depth"] += 1
        try:
            return fn(*a, **k)
        finally:
            H._HEAD_PRECISION["depth"] -= 1
    wrapped.__name__ = getattr(fn, "__name__", "wrapped")
    return wrapped


for _cls in [v for v in list(globals().values()) if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function]:
    _cls.forward = staticmethod(_exact(_cls.forward))
    _cls.backward = staticmethod(_exact(_cls.backward))
