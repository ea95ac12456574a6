"""Training-mode BatchNorm + activation and squeeze-and-excitation of the MBConv / U-Net decoder blocks on libsrbh
(csrc/srbh_mbconv.hip), with hand-written backward.

The reference runs these blocks through segmentation_models_pytorch (mymodels.py:242-258, forward at mymodels.py:276-287); in
training each MBConv block is ~14 stock launches forward and ~30 backward around its three convolutions.  Here

* ``bn_act_train(bn, x, act, res, drop)``   = F.batch_norm(training=True) -> SiLU / ReLU / none -> [* drop-connect factor] -> [+ skip]
  in ONE launch forward and ONE backward;
* ``bn_swish_se_train(bn, x, se_reduce, se_expand)`` = BatchNorm -> SiLU -> avg-pool -> 1x1 reduce -> SiLU -> 1x1 expand -> sigmoid ->
  scale in THREE launches forward and FOUR backward (the plane means come out of the BatchNorm kernel; the backward of the
  gate re-enters the BatchNorm backward kernel as a per-plane scale and offset of dy).

Both update ``running_mean`` / ``running_var`` exactly as nn.BatchNorm2d does and count the batch through hrfuse.note_batch.  Shapes
the kernels do not take (plane sizes that are not a multiple of 4, SyncBatchNorm, CPU tensors) return None from ``supported`` and the caller
keeps the stock ops.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
from torch import nn

from . import _lib, wcache

ENABLED = os.environ.get("SRBH_ENC_TRAIN_FUSED", "1") == "1"
_ACT = {None: 0, "silu": 1, "relu": 2}


def supported(bn, x) -> bool:
    if not (ENABLED and bn.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and type(bn) is nn.BatchNorm2d
            and bn.affine and bn.track_running_stats and bn.momentum is not None and torch.is_grad_enabled()):
        return False
    B, Cc, H, W = x.shape
    return bool(_lib.lib().srbh_bn_act_train_supported(B, Cc, H * W))


def _ptr(t):
    return None if t is None else t.data_ptr()


def _note(bn):
    from . import hrfuse as _H
    _H.note_batch(bn)
    wcache.stamp((bn.running_mean, bn.running_var))          # written by the kernel, behind the version counters
    if _H._NBT["depth"] == 0:
        _H.flush_batches()


def _ws(x):
    B, Cc, H, W = x.shape
    n = _lib.lib().srbh_bn_act_train_ws_bytes(B, Cc, H * W)
    return torch.empty(n // 8, dtype=torch.float64, device=x.device) if n else None


def _fwd(x, bn, act, y, mean, invstd, pooled=None, res=None, drop=None):
    B, Cc, H, W = x.shape
    ws = _ws(x)
    a = _lib.BnActArgs(ws=_ptr(ws), x=x.data_ptr(), y=y.data_ptr(), gamma=bn.weight.data_ptr(), beta=bn.bias.data_ptr(),
                       running_mean=bn.running_mean.data_ptr(), running_var=bn.running_var.data_ptr(), save_mean=mean.data_ptr(),
                       save_invstd=invstd.data_ptr(), pooled=_ptr(pooled), res=_ptr(res), drop=_ptr(drop), momentum=float(bn.momentum),
                       eps=float(bn.eps), B=B, C=Cc, HW=H * W, act=act)
    _lib.check(_lib.lib().srbh_bn_act_train_fwd(C.byref(a), _lib.stream_ptr()), "bn_act_train_fwd")


def _bwd(dy, x, gamma, beta, mean, invstd, act, need_dx, gate=None, dpooled=None, drop=None):
    B, Cc, H, W = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dgamma = torch.empty_like(gamma)
    dbeta = torch.empty_like(beta)
    ws = _ws(x)
    a = _lib.BnActBwdArgs(ws=_ptr(ws), dy=dy.data_ptr(), x=x.data_ptr(), gamma=gamma.data_ptr(), beta=beta.data_ptr(), save_mean=mean.data_ptr(),
                          save_invstd=invstd.data_ptr(), gate=_ptr(gate), dpooled=_ptr(dpooled), drop=_ptr(drop), dx=_ptr(dx),
                          dgamma=dgamma.data_ptr(), dbeta=dbeta.data_ptr(), B=B, C=Cc, HW=H * W, act=act)
    _lib.check(_lib.lib().srbh_bn_act_train_bwd(C.byref(a), _lib.stream_ptr()), "bn_act_train_bwd")
    return dx, dgamma, dbeta


class _BnActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, res, drop, bn, act):
        x = x.contiguous()
        if res is not None:
            res = res.contiguous()
        y = torch.empty_like(x)
        Cc = x.shape[1]
        mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        _fwd(x, bn, act, y, mean, invstd, None, res, drop)
        ctx.save_for_backward(x, gamma, beta, mean, invstd, drop)
        ctx.act = act
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd, drop = ctx.saved_tensors
        dy = dy.contiguous()
        dx, dgamma, dbeta = _bwd(dy, x, gamma, beta, mean, invstd, ctx.act, ctx.needs_input_grad[0], drop=drop)
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, dgamma, dbeta, dres, None, None, None


class _BnSwishSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, w1, b1, w2, b2, bn):
        x = x.contiguous()
        B, Cc, H, W = x.shape
        SQ = w1.shape[0]
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty(Cc, dtype=torch.float32, device=dev)
        invstd = torch.empty_like(mean)
        small = torch.empty(2 * B * Cc + 2 * B * SQ, dtype=torch.float32, device=dev)     # pooled | gate | hidden | hidden_pre
        pooled, gate = small[:B * Cc], small[B * Cc:2 * B * Cc]
        hidden, hidden_pre = small[2 * B * Cc:2 * B * Cc + B * SQ], small[2 * B * Cc + B * SQ:]
        _fwd(x, bn, 1, y, mean, invstd, pooled)
        _lib.check(_lib.lib().srbh_se_train_fwd(y.data_ptr(), pooled.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                                hidden.data_ptr(), hidden_pre.data_ptr(), gate.data_ptr(), B, Cc, SQ, H * W,
                                                _lib.stream_ptr()), "se_train_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, invstd, small, w1, w2)
        ctx.geo = (B, Cc, SQ, H * W)
        return y

    @staticmethod
    def backward(ctx, dout):
        x, gamma, beta, mean, invstd, small, w1, w2 = ctx.saved_tensors
        B, Cc, SQ, HW = ctx.geo
        dev = x.device
        dout = dout.contiguous()
        pooled, gate = small[:B * Cc], small[B * Cc:2 * B * Cc]
        hidden, hidden_pre = small[2 * B * Cc:2 * B * Cc + B * SQ], small[2 * B * Cc + B * SQ:]
        L = _lib.lib()
        ws = torch.empty(L.srbh_se_train_bwd_ws_floats(B, Cc, SQ) + B * Cc, dtype=torch.float32, device=dev)
        dpooled = ws[-B * Cc:]
        dw1 = torch.empty_like(w1)
        dw2 = torch.empty_like(w2)
        db1 = torch.empty(SQ, dtype=torch.float32, device=dev)
        db2 = torch.empty(Cc, dtype=torch.float32, device=dev)
        _lib.check(L.srbh_se_train_bwd(dout.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                       pooled.data_ptr(), hidden.data_ptr(), hidden_pre.data_ptr(), gate.data_ptr(), w1.data_ptr(),
                                       w2.data_ptr(), ws.data_ptr(), dpooled.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(),
                                       db2.data_ptr(), B, Cc, SQ, HW, 1, _lib.stream_ptr()), "se_train_bwd")
        dx, dgamma, dbeta = _bwd(dout, x, gamma, beta, mean, invstd, 1, ctx.needs_input_grad[0], gate=gate, dpooled=dpooled)
        return dx, dgamma, dbeta, dw1, db1, dw2, db2, None


MID_FUSED = os.environ.get("SRBH_MBCONV_MID", "1") == "1"


def mid_supported(bn0, bn1, dw, x) -> bool:
    """the fused MBConv middle (BatchNorm0 + SiLU -> depthwise -> BatchNorm1 + SiLU + pool, one launch each way) takes this block"""
    if not (MID_FUSED and supported(bn0, x) and type(bn1) is nn.BatchNorm2d and bn1.training and bn1.affine and bn1.track_running_stats
            and bn1.momentum is not None and getattr(dw, "_depthwise", False) and dw.weight.dtype == torch.float32):
        return False
    B, Cc, H, W = x.shape
    pl, pr, pt, pb = dw._pad
    K = dw.weight.shape[-1]
    return (dw.stride[0] == 1 and pl == pr == pt == pb == K // 2 and bn1.num_features == Cc
            and bool(_lib.lib().srbh_mbconv_mid_supported(B, Cc, H, W, K, 1)))


class _MidSEFn(torch.autograd.Function):
    """BatchNorm0 + SiLU -> depthwise conv -> BatchNorm1 + SiLU -> squeeze-and-excitation of one MBConv block (efficientnet_pytorch
    MBConvBlock.forward): srbh_mbconv_mid_fwd + srbh_se_train_fwd forward, srbh_se_train_bwd + srbh_mbconv_mid_bwd backward -- the
    separate Functions ran bn_act, depthwise, bn_act + pool, squeeze-excite (4 + 2 launches) and their backwards (4 + ~5)."""

    @staticmethod
    def forward(ctx, e_pre, g0, b0, wdw, g1, b1, w1, bb1, w2, bb2, bn0, bn1):
        e_pre = e_pre.contiguous()
        wdw = wdw.contiguous()
        B, Cc, H, W = e_pre.shape
        SQ = w1.shape[0]
        K = wdw.shape[-1]
        dev = e_pre.device
        d_pre = torch.empty_like(e_pre)
        y = torch.empty_like(e_pre)
        stats = torch.empty(4 * Cc, dtype=torch.float32, device=dev)           # mean0 | invstd0 | mean1 | invstd1
        small = torch.empty(2 * B * Cc + 2 * B * SQ, dtype=torch.float32, device=dev)     # pooled | gate | hidden | hidden_pre
        pooled, gate = small[:B * Cc], small[B * Cc:2 * B * Cc]
        hidden, hidden_pre = small[2 * B * Cc:2 * B * Cc + B * SQ], small[2 * B * Cc + B * SQ:]
        sp = stats.data_ptr()
        a = _lib.MbMidArgs(e_pre=e_pre.data_ptr(), wdw=wdw.data_ptr(), gamma0=g0.data_ptr(), beta0=b0.data_ptr(),
                           running_mean0=bn0.running_mean.data_ptr(), running_var0=bn0.running_var.data_ptr(), mean0=sp, invstd0=sp + 4 * Cc,
                           gamma1=g1.data_ptr(), beta1=b1.data_ptr(), running_mean1=bn1.running_mean.data_ptr(),
                           running_var1=bn1.running_var.data_ptr(), mean1=sp + 8 * Cc, invstd1=sp + 12 * Cc, d_pre=d_pre.data_ptr(),
                           y=y.data_ptr(), pooled=pooled.data_ptr(), momentum0=float(bn0.momentum), eps0=float(bn0.eps),
                           momentum1=float(bn1.momentum), eps1=float(bn1.eps), B=B, C=Cc, H=H, W=W, K=K)
        L = _lib.lib()
        _lib.check(L.srbh_mbconv_mid_fwd(C.byref(a), _lib.stream_ptr()), "mbconv_mid_fwd")
        _lib.check(L.srbh_se_train_fwd(y.data_ptr(), pooled.data_ptr(), w1.data_ptr(), bb1.data_ptr(), w2.data_ptr(), bb2.data_ptr(),
                                       hidden.data_ptr(), hidden_pre.data_ptr(), gate.data_ptr(), B, Cc, SQ, H * W, _lib.stream_ptr()), "se_train_fwd")
        ctx.save_for_backward(e_pre, d_pre, g0, b0, wdw, g1, b1, stats, small, w1, w2)
        ctx.geo = (B, Cc, SQ, H, W, K)
        return y

    @staticmethod
    def backward(ctx, dout):
        e_pre, d_pre, g0, b0, wdw, g1, b1, stats, small, w1, w2 = ctx.saved_tensors
        B, Cc, SQ, H, W, K = ctx.geo
        HW = H * W
        dev = e_pre.device
        dout = dout.contiguous()
        mean0, invstd0, mean1, invstd1 = stats[:Cc], stats[Cc:2 * Cc], stats[2 * Cc:3 * Cc], stats[3 * Cc:]
        pooled, gate = small[:B * Cc], small[B * Cc:2 * B * Cc]
        hidden, hidden_pre = small[2 * B * Cc:2 * B * Cc + B * SQ], small[2 * B * Cc + B * SQ:]
        L = _lib.lib()
        ws = torch.empty(L.srbh_se_train_bwd_ws_floats(B, Cc, SQ) + B * Cc, dtype=torch.float32, device=dev)
        dpooled = ws[-B * Cc:]
        dw1 = torch.empty_like(w1)
        dw2 = torch.empty_like(w2)
        db1 = torch.empty(SQ, dtype=torch.float32, device=dev)
        db2 = torch.empty(Cc, dtype=torch.float32, device=dev)
        _lib.check(L.srbh_se_train_bwd(dout.data_ptr(), d_pre.data_ptr(), g1.data_ptr(), b1.data_ptr(), mean1.data_ptr(), invstd1.data_ptr(),
                                       pooled.data_ptr(), hidden.data_ptr(), hidden_pre.data_ptr(), gate.data_ptr(), w1.data_ptr(),
                                       w2.data_ptr(), ws.data_ptr(), dpooled.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(),
                                       db2.data_ptr(), B, Cc, SQ, HW, 1, _lib.stream_ptr()), "se_train_bwd")
        de = torch.empty_like(e_pre) if ctx.needs_input_grad[0] else None
        dwdw = torch.empty_like(wdw)
        aff = torch.empty(4 * Cc, dtype=torch.float32, device=dev)            # dgamma0 | dbeta0 | dgamma1 | dbeta1
        ap = aff.data_ptr()
        a = _lib.MbMidBwdArgs(dout=dout.data_ptr(), gate=gate.data_ptr(), dpooled=dpooled.data_ptr(), d_pre=d_pre.data_ptr(),
                              e_pre=e_pre.data_ptr(), wdw=wdw.data_ptr(), gamma0=g0.data_ptr(), beta0=b0.data_ptr(), mean0=mean0.data_ptr(),
                              invstd0=invstd0.data_ptr(), gamma1=g1.data_ptr(), beta1=b1.data_ptr(), mean1=mean1.data_ptr(),
                              invstd1=invstd1.data_ptr(), de_pre=_ptr(de), dwdw=dwdw.data_ptr(), dgamma0=ap, dbeta0=ap + 4 * Cc,
                              dgamma1=ap + 8 * Cc, dbeta1=ap + 12 * Cc, B=B, C=Cc, H=H, W=W, K=K)
        _lib.check(L.srbh_mbconv_mid_bwd(C.byref(a), _lib.stream_ptr()), "mbconv_mid_bwd")
        return (de, aff[:Cc], aff[Cc:2 * Cc], dwdw, aff[2 * Cc:3 * Cc], aff[3 * Cc:], dw1, db1, dw2, db2, None, None)


def mid_se_train(bn0, e_pre, dw, bn1, se_reduce, se_expand):
    """swish(bn0(e_pre)) -> depthwise conv -> swish(bn1(.)) -> squeeze-and-excitation, training mode, as two launches forward and two
    backward (plus squeeze-excite's own small kernels)"""
    y = _MidSEFn.apply(e_pre, bn0.weight, bn0.bias, dw.weight, bn1.weight, bn1.bias, se_reduce.weight, se_reduce.bias,
                       se_expand.weight, se_expand.bias, bn0, bn1)
    _note(bn0)
    _note(bn1)
    return y


def bn_act_train(bn, x, act=None, res=None, drop=None):
    """y = act(batch_norm(x)) [* drop[b]] [+ res]; `drop`: contiguous (B,) factors or None"""
    y = _BnActFn.apply(x, bn.weight, bn.bias, res, drop, bn, _ACT[act])
    _note(bn)
    return y


def se_supported(se_reduce, se_expand) -> bool:
    return (se_reduce.weight.is_contiguous() and se_expand.weight.is_contiguous() and se_reduce.bias is not None
            and se_expand.bias is not None and se_reduce.weight.shape[0] <= 256)


def bn_swish_se_train(bn, x, se_reduce, se_expand):
    """sigmoid(se_expand(swish(se_reduce(avg_pool(s))))) * s with s = swish(batch_norm(x)) (efficientnet_pytorch MBConvBlock.forward)"""
    y = _BnSwishSEFn.apply(x, bn.weight, bn.bias, se_reduce.weight, se_reduce.bias, se_expand.weight, se_expand.bias, bn)
    _note(bn)
    return y


class _Up2CatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, skip):
        x = x.contiguous()
        B, Cx, H, W = x.shape
        Cs = 0
        if skip is not None:
            skip = skip.contiguous()
            Cs = skip.shape[1]
        out = torch.empty((B, Cx + Cs, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().srbh_up2_cat_fwd(x.data_ptr(), _ptr(skip), out.data_ptr(), B, Cx, Cs, H, W, _lib.stream_ptr()), "up2_cat_fwd")
        ctx.geo = (B, Cx, Cs, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, Cx, Cs, H, W = ctx.geo
        dout = dout.contiguous()
        dx = torch.empty((B, Cx, H, W), dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[0] else None
        dskip = (torch.empty((B, Cs, 2 * H, 2 * W), dtype=torch.float32, device=dout.device)
                 if (Cs and ctx.needs_input_grad[1]) else None)
        _lib.check(_lib.lib().srbh_up2_cat_bwd(dout.data_ptr(), _ptr(dx), _ptr(dskip), B, Cx, Cs, H, W, _lib.stream_ptr()), "up2_cat_bwd")
        return dx, dskip


def up2_cat_supported(x, skip) -> bool:
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[-1] % 2 == 0
            and (skip is None or (skip.is_cuda and skip.dtype == torch.float32 and skip.shape[0] == x.shape[0]
                                  and skip.shape[2] == 2 * x.shape[2] and skip.shape[3] == 2 * x.shape[3])))


def up2_cat(x, skip=None):
    """torch.cat([F.interpolate(x, scale_factor=2, mode="nearest"), skip], dim=1) (smp DecoderBlock.forward), one launch each way"""
    return _Up2CatFn.apply(x, skip)
