"""Third-party parts of the height model, restated in stock PyTorch(-ROCm) ops.

The reference builds its encoder / decoders from ``segmentation_models_pytorch`` (unpinned in requirements.txt:14;
call sites mymodels.py:242-258,276,279,287), which is not in the reference tree and not installable offline.
SURVEY.md 8(a) a18 scopes these < 1 GFLOP/tile parts as "stock PyTorch-ROCm ops acceptable"; this file restates
the published architectures from their papers / documented layouts:

* ``EfficientNetEncoder``  -- EfficientNet (Tan & Le 2019) B0-B7 feature extractor with the module/parameter naming
  of efficientnet_pytorch 0.7.1 as wrapped by smp <0.5 (``_conv_stem``, ``_bn0``, ``_blocks.{i}._expand_conv`` ...,
  ``_conv_head``/``_bn1`` present but unused, ``_fc`` removed), TF-style "same" padding computed statically for the
  model's nominal resolution, BatchNorm eps 1e-3 / momentum 0.01, swish, squeeze-excite 0.25, drop-connect.
* ``UnetDecoder``          -- U-Net decoder of smp <0.5: n_blocks x [nearest x2, concat skip, (conv3x3+BN+ReLU) x2],
  keys ``blocks.{i}.conv{1,2}.{0,1}.*``.

PARITY UNPINNED: the reference holds no fixture at this boundary and the dependency is absent; the anchors are the
author's recorded parameter counts (mymodels.py:765: encoder 17.55 M, decoder 2.68 M), asserted in tests.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn.functional as F
from . import wcache
from torch import nn

# (width, depth, resolution, dropout) -- EfficientNet paper table / efficientnet_pytorch.utils.efficientnet_params
_EFFNET_PARAMS = {
    "efficientnet-b0": (1.0, 1.0, 224, 0.2), "efficientnet-b1": (1.0, 1.1, 240, 0.2),
    "efficientnet-b2": (1.1, 1.2, 260, 0.3), "efficientnet-b3": (1.2, 1.4, 300, 0.3),
    "efficientnet-b4": (1.4, 1.8, 380, 0.4), "efficientnet-b5": (1.6, 2.2, 456, 0.4),
    "efficientnet-b6": (1.8, 2.6, 528, 0.5), "efficientnet-b7": (2.0, 3.1, 600, 0.5),
}
# (repeats, kernel, stride, expand, in, out) of the B0 baseline; SE ratio 0.25 everywhere
_BASE_BLOCKS = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
                (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
# smp encoder registry: stage boundaries (block indices) and feature channels
_SMP_STAGES = {
    "efficientnet-b0": ((3, 5, 9, 16), (3, 32, 24, 40, 112, 320)), "efficientnet-b1": ((5, 8, 16, 23), (3, 32, 24, 40, 112, 320)),
    "efficientnet-b2": ((5, 8, 16, 23), (3, 32, 24, 48, 120, 352)), "efficientnet-b3": ((5, 8, 18, 26), (3, 40, 32, 48, 136, 384)),
    "efficientnet-b4": ((6, 10, 22, 32), (3, 48, 32, 56, 160, 448)), "efficientnet-b5": ((8, 13, 27, 39), (3, 48, 40, 64, 176, 512)),
    "efficientnet-b6": ((9, 15, 31, 45), (3, 56, 40, 72, 200, 576)), "efficientnet-b7": ((11, 18, 38, 55), (3, 64, 48, 80, 224, 640)),
}
BN_MOM, BN_EPS = 0.01, 1e-3
DROP_CONNECT = 0.2

# "no silent fallback".  a18 may run stock ops, but which call sites of a DEVICE forward took them is counted
# here -- key (site, shape) -> calls -- so that a hot shape sliding back to MIOpen / ATen shows up as a count (bench.py prints
# `encdec_kernels.stock_ops`), not only as time.  CPU tensors are the oracle-side path and are not counted.
STOCK_OPS: dict = {}


def _stock(site, x):
    if x.is_cuda:
        k = (site, tuple(x.shape[1:]))
        STOCK_OPS[k] = STOCK_OPS.get(k, 0) + 1


def stock_ops_reset():
    STOCK_OPS.clear()


def stock_ops_summary(top=6):
    """{'calls': total, 'by_site': {site: calls}, 'top': [[site, shape, calls], ...]} of the device forwards since the last reset"""
    by = {}
    for (site, _), n in STOCK_OPS.items():
        by[site] = by.get(site, 0) + n
    rows = sorted(STOCK_OPS.items(), key=lambda kv: -kv[1])[:top]
    return {"calls": sum(STOCK_OPS.values()), "by_site": dict(sorted(by.items(), key=lambda kv: -kv[1])),
            "top": [[s, list(shape), n] for (s, shape), n in rows]}


def _round_filters(f, width, divisor=8):
    f *= width
    new = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if new < 0.9 * f:
        new += divisor
    return int(new)


def _round_repeats(r, depth):
    return int(math.ceil(depth * r))


class _DepthwiseConvFn(torch.autograd.Function):
    """Depthwise KxK conv (K 3|5, stride 1|2) with the static "same" zero padding folded in, on libsrbh (csrc/srbh_dwconv.hip):
    MIOpen serves these fp32 shapes with its naive kernels.  Device fp32 NCHW tensors only (the CPU path of the encoder
    stays on stock ops)."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        from . import _lib
        pl, pr, pt, pb = pad
        x = x.contiguous()
        weight = weight.contiguous()
        B, C, H, W = x.shape
        K = weight.shape[-1]
        OH, OW = (H + pt + pb - K) // stride + 1, (W + pl + pr - K) // stride + 1
        y = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().srbh_dwconv_fwd(x.data_ptr(), weight.data_ptr(), y.data_ptr(), B, C, H, W, K, stride, pt, pl, OH, OW,
                                              _lib.stream_ptr()), "dwconv_fwd")
        ctx.save_for_backward(x, weight)
        ctx.geo = (B, C, H, W, K, stride, pt, pl, OH, OW)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        x, weight = ctx.saved_tensors
        B, C, H, W, K, stride, pt, pl, OH, OW = ctx.geo
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(_lib.lib().srbh_dwconv_bwd_data(dy.data_ptr(), weight.data_ptr(), dx.data_ptr(), B, C, H, W, K, stride, pt, pl,
                                                       OH, OW, _lib.stream_ptr()), "dwconv_bwd_data")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            ws = torch.empty(_lib.lib().srbh_dwconv_bwd_weight_splits(B, C) * C * K * K, dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().srbh_dwconv_bwd_weight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, C, H, W, K,
                                                         stride, pt, pl, OH, OW, _lib.stream_ptr()), "dwconv_bwd_weight")
        return dx, dw, None, None


# the skip connection of an MBConv block taken through its expand conv's autograd node (see _PointwiseConvFn.forward); SRBH_SKIP_EXPAND=0: autograd's add
SKIP_THROUGH_EXPAND = os.environ.get("SRBH_SKIP_EXPAND", "1") == "1"

class _PointwiseConvFn(torch.autograd.Function):
    """1x1 convolution without bias (MBConv expand / project) on libsrbh (csrc/srbh_pwconv.hip): one fp32-MFMA launch forward, one for
    the input gradient, one (+ an ordered reduce when it splits over images) for the weight gradient.  MIOpen wraps its NHWC kernels
    for these shapes in batched transposes, zero fills and split-K atomics: ~390 launches / 4.1 ms of the training step."""

    @staticmethod
    def forward(ctx, x, weight, wt=None, skip=False):
        """wt: the transposed weight [Cin][Cout] made by `PointwiseTransposes` for THIS state of `weight` (or None).
        skip: also return x itself (an alias) -- the block's skip connection taken THROUGH this node, so that in backward the gradient arriving
        over the skip meets the conv's input gradient inside one kernel (srbh_pwconv_bwd_data_res) instead of in an add launched by autograd"""
        from . import _lib
        x = x.contiguous()
        weight = weight.contiguous()
        ctx.skip = bool(skip)
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
        if wt is not None:
            _lib.check(_lib.lib().srbh_pwconv_fwd_wt(x.data_ptr(), wt.data_ptr(), y.data_ptr(), B, Cin, Cout, H * W, _lib.stream_ptr()), "pwconv_fwd_wt")
        else:
            _lib.check(_lib.lib().srbh_pwconv_fwd(x.data_ptr(), weight.data_ptr(), y.data_ptr(), B, Cin, Cout, H * W, _lib.stream_ptr()), "pwconv_fwd")
        ctx.save_for_backward(x, weight)
        if ctx.skip:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        from . import _lib
        x, weight = ctx.saved_tensors
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dy = dy.contiguous()
        L = _lib.lib()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if dskip is not None:
                dskip = dskip.contiguous()
                _lib.check(L.srbh_pwconv_bwd_data_res(dy.data_ptr(), weight.data_ptr(), dskip.data_ptr(), dx.data_ptr(), B, Cin, Cout, H * W,
                                                      _lib.stream_ptr()), "pwconv_bwd_data_res")
            else:
                _lib.check(L.srbh_pwconv_bwd_data(dy.data_ptr(), weight.data_ptr(), dx.data_ptr(), B, Cin, Cout, H * W, _lib.stream_ptr()), "pwconv_bwd_data")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            n = L.srbh_pwconv_bwd_weight_ws_floats(B, Cin, Cout, H * W)
            ws = torch.empty(n, dtype=torch.float32, device=x.device) if n else None
            _lib.check(L.srbh_pwconv_bwd_weight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr() if n else None, B, Cin, Cout, H * W,
                                                _lib.stream_ptr()), "pwconv_bwd_weight")
        return dx, dw, None, None


class PointwiseTransposes:
    """W^T [Cin][Cout] of every 1x1 convolution of an encoder in ONE flat buffer, refreshed by ONE launch (srbh_transpose_many) at the
    start of a forward whenever a weight changed -- in training that is once per step.  The forward kernel then reads the weights as
    the input-gradient kernel reads W: coalesced along the output channel."""

    def __init__(self, convs):
        self.convs = list(convs)
        self.key = None
        self.flat = None
        self.table = None
        self.table_ptrs = None

    def _key(self):
        return tuple((c.weight._version, c.weight.data_ptr()) for c in self.convs) + wcache.gen(*[c.weight for c in self.convs])

    def refresh(self, device):
        from . import _lib
        import numpy as np
        key = (self._key(), str(device))
        if key != self.key:
            if self.flat is None or self.flat.device != device or self.table_ptrs != tuple(c.weight.data_ptr() for c in self.convs):
                self.flat = torch.empty(sum(c.weight.numel() for c in self.convs), dtype=torch.float32, device=device)
                desc = np.zeros(len(self.convs), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4")]))
                o = 0
                for i, c in enumerate(self.convs):
                    cout, cin = c.weight.shape[:2]
                    c.__dict__["_srbh_wt"] = self.flat[o:o + cout * cin].view(cin, cout)
                    desc[i] = (c.weight.data_ptr(), self.flat.data_ptr() + 4 * o, cout, cin)
                    o += cout * cin
                self.table = torch.from_numpy(desc.view(np.uint8).copy()).to(device)
                self.table_ptrs = tuple(c.weight.data_ptr() for c in self.convs)
            _lib.check(_lib.lib().srbh_transpose_many(self.table.data_ptr(), len(self.convs), _lib.stream_ptr()), "transpose_many")
            self.key = key
            for c in self.convs:
                c.__dict__["_srbh_wt_state"] = (c.weight._version, c.weight.data_ptr(), wcache.gen(c.weight))
        wcache.keep(self.flat, self.table)


PWCONV = __import__("os").environ.get("SRBH_PWCONV", "1")      # "1" (default): training and inference (+0.8 % tiled predict, -0.7 ms train step vs MIOpen); "train": only when gradients are recorded; "0": never

# (history, round 2: restating the 1x1 convs as batched rocBLAS GEMMs -- forward W @ X_b, data gradient
# W^T @ dY_b, weight gradient sum_b dY_b @ X_b^T -- removed ~220 of the step's launches (MIOpen's backward wraps its NHWC
# implicit-GEMM kernels in batched transposes and zero fills) but the step got 1.5 ms SLOWER: at these shapes the Tensile kernels
# behind bmm lose more than the launches cost.  Measured round 2)
_ACT = {None: 0, "silu": 1, "relu": 2}
FUSED_SE_EVAL = True       # squeeze-and-excitation of the MBConv blocks as three libsrbh launches at inference
FUSED_BN_EVAL = True      # tests switch it off to compare with the stock inference-BatchNorm path
# MBConv block at inference as 4-5 launches (round 4): expand 1x1 | bn0+swish -> depthwise -> bn1+swish -> pool in ONE kernel | SE hidden |
# SE gate | project 1x1 with the gate on its operand and bn2 (+ skip) in its epilogue -- instead of 8 launches and ten passes over the
# 6x-expanded tensor.  SRBH_MBCONV_EVAL=0: the separate launches (A/B aid, and what the parity tests compare against).
MBCONV_EVAL = __import__("os").environ.get("SRBH_MBCONV_EVAL", "1") == "1"


def bn_act(bn, x, act=None, res=None, drop=None):
    """BatchNorm2d followed by an activation [, the drop-connect factor `drop` (B,) and the skip connection `res`].  Inference on
    the device: ONE libsrbh pass y = act(x * scale + shift) with the running statistics folded into (scale, shift)
    (csrc/srbh_dwconv.hip) -- MIOpen's inference-BatchNorm kernel costs ~39 us per call whatever the size, 8.6 % of the
    tiled-inference path.  Training on the device, planes below 32x32: one libsrbh launch forward and one backward
    (mbconv_autograd.bn_act_train).  Otherwise (CPU, large planes, SyncBatchNorm): the stock ops."""
    if x.is_cuda and bn.training:
        from . import mbconv_autograd as MB
        if MB.supported(bn, x):
            return MB.bn_act_train(bn, x, act, res, drop)
    if _fused_eval_ok(bn, x):
        from . import _lib
        scale, shift = _bn_affine(bn, x.device)
        x = x.contiguous()
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        if res is not None:        # the block's skip connection in the same pass (res: what the caller would add to the result)
            res = res.contiguous()
            _lib.check(_lib.lib().srbh_affine_act_add_nchw(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), res.data_ptr(), y.data_ptr(),
                                                           B, C, H * W, _ACT[act], _lib.stream_ptr()), "affine_act_add_nchw")
            return y
        _lib.check(_lib.lib().srbh_affine_act_nchw(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), B, C, H * W,
                                                   _ACT[act], _lib.stream_ptr()), "affine_act_nchw")
        return y
    _stock("batch_norm_train" if bn.training else "batch_norm_eval", x)
    if bn.training and type(bn) is nn.BatchNorm2d and bn.track_running_stats and bn.momentum is not None:
        # the module's own forward, minus its per-module `num_batches_tracked.add_(1)` launch (hrfuse.note_batch: one fused
        # increment per model forward); SyncBatchNorm and exotic configurations keep the module call
        from . import hrfuse as _H
        _H.note_batch(bn)
        x = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps)
        if _H._NBT["depth"] == 0:
            _H.flush_batches()
    else:
        x = bn(x)
    if act == "silu":
        x = _swish(x)
    elif act == "relu":
        x = F.relu(x)
    if drop is not None:
        x = x * drop.view(-1, 1, 1, 1)
    return x if res is None else x + res


def _bn_affine(bn, device):
    """(scale, shift) of an inference BatchNorm, cached on the module and invalidated by parameter / buffer versions"""
    from . import _lib
    C = bn.num_features
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, bn.weight.data_ptr(),
           wcache.gen(bn.weight, bn.bias, bn.running_mean, bn.running_var))     # (buffers are stamped by graph replays: harness.TrainStep)
    cache = bn.__dict__.get("_srbh_affine")
    if cache is None or cache[0] != key:
        scale = torch.empty(C, dtype=torch.float32, device=device)
        shift = torch.empty(C, dtype=torch.float32, device=device)
        _lib.check(_lib.lib().srbh_bn_eval_scale_shift(C, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                                       bn.running_var.data_ptr(), bn.eps, scale.data_ptr(), shift.data_ptr(),
                                                       _lib.stream_ptr()), "bn_eval_scale_shift")
        cache = (key, scale, shift)
        bn.__dict__["_srbh_affine"] = cache
    wcache.keep(cache)
    return cache[1], cache[2]


def _fused_eval_ok(bn, x):
    return (FUSED_BN_EVAL and (not bn.training) and x.is_cuda and x.dtype == torch.float32 and bn.track_running_stats
            and not torch.is_grad_enabled())


# the stem at inference as one libsrbh pass (conv3x3 stride 2 + folded BatchNorm + SiLU: srbh_stem_conv_eval); SRBH_STEM_EVAL=0: MIOpen's conv + the affine pass
STEM_EVAL = os.environ.get("SRBH_STEM_EVAL", "1") == "1"


def _stem_eval_ok(conv, bn, x):
    if not (STEM_EVAL and _fused_eval_ok(bn, x) and x.dim() == 4 and conv.bias is None and conv.groups == 1 and conv.weight.dtype == torch.float32
            and conv.kernel_size == (3, 3) and conv.stride[0] == conv.stride[1] and conv.dilation == (1, 1)):
        return False
    from . import _lib
    return bool(_lib.lib().srbh_stem_conv_eval_supported(conv.in_channels, conv.out_channels, 3))


def _stem_eval(conv, bn, x):
    from . import _lib
    scale, shift = _bn_affine(bn, x.device)
    x = x.contiguous()
    B, Cin, H, W = x.shape
    pl, pr, pt, pb = conv._pad
    st = conv.stride[0]
    OH, OW = (H + pt + pb - 3) // st + 1, (W + pl + pr - 3) // st + 1
    y = torch.empty((B, conv.out_channels, OH, OW), dtype=torch.float32, device=x.device)
    w = conv.weight.detach().contiguous()
    _lib.check(_lib.lib().srbh_stem_conv_eval(x.data_ptr(), w.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), B, Cin, H, W,
                                              conv.out_channels, st, pt, pl, OH, OW, _ACT["silu"], _lib.stream_ptr()), "stem_conv_eval")
    return y


def bn_swish_se(bn, x, se_reduce, se_expand):
    """MBConv middle at inference: swish(bn(x)) followed by squeeze-and-excitation, as THREE libsrbh launches
    (csrc/srbh_dwconv.hip) instead of ~11 stock-op ones per block: the encoder is bound by its launch count."""
    from . import _lib
    L = _lib.lib()
    scale, shift = _bn_affine(bn, x.device)
    x = x.contiguous()
    B, C, H, W = x.shape
    SQ = se_reduce.out_channels
    y = torch.empty_like(x)
    pooled = torch.empty((B, C), dtype=torch.float32, device=x.device)
    _lib.check(L.srbh_affine_act_pool_nchw(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), pooled.data_ptr(), B, C,
                                           H * W, 1, _lib.stream_ptr()), "affine_act_pool_nchw")
    hidden = torch.empty((B, SQ), dtype=torch.float32, device=x.device)
    _lib.check(L.srbh_se_hidden(pooled.data_ptr(), se_reduce.weight.data_ptr(), se_reduce.bias.data_ptr(), hidden.data_ptr(), B, C, SQ,
                                _lib.stream_ptr()), "se_hidden")
    _lib.check(L.srbh_se_gate_scale(y.data_ptr(), hidden.data_ptr(), se_expand.weight.data_ptr(), se_expand.bias.data_ptr(), B, C, SQ,
                                    H * W, _lib.stream_ptr()), "se_gate_scale")
    return y


class SamePadConv2d(nn.Conv2d):
    """Conv2d with TensorFlow "same" padding fixed at construction for a nominal input size (the padding is a
    parameter-free ``static_padding`` sub-module, so state_dict keys are just weight/bias)."""

    def __init__(self, in_ch, out_ch, kernel_size, image_size, stride=1, groups=1, bias=True):
        super().__init__(in_ch, out_ch, kernel_size, stride=stride, groups=groups, bias=bias)
        ih = iw = image_size
        kh, kw = self.weight.shape[-2:]
        sh, sw = self.stride
        oh, ow = math.ceil(ih / sh), math.ceil(iw / sw)
        pad_h = max((oh - 1) * sh + (kh - 1) + 1 - ih, 0)
        pad_w = max((ow - 1) * sw + (kw - 1) + 1 - iw, 0)
        self._pad = (pad_w // 2, pad_w - pad_w // 2, pad_h // 2, pad_h - pad_h // 2)   # (left, right, top, bottom)
        if pad_h > 0 or pad_w > 0:
            self.static_padding = nn.ZeroPad2d(self._pad)
        else:
            self.static_padding = nn.Identity()
        self._pointwise = (kh == kw == 1 and sh == sw == 1 and groups == 1 and not bias and pad_h == 0 and pad_w == 0)
        self._depthwise = (groups == in_ch == out_ch and groups > 1 and kh == kw and kh in (3, 5) and sh == sw and sh in (1, 2)
                           and not bias)

    def forward(self, x):
        if self._depthwise and x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32:
            return _DepthwiseConvFn.apply(x, self.weight, self.stride[0], self._pad)
        if (self._pointwise and x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32 and x.dim() == 4
                and (PWCONV == "1" or (PWCONV == "train" and torch.is_grad_enabled()))):
            from . import _lib
            if _lib.lib().srbh_pwconv_supported(x.shape[0], x.shape[1], self.weight.shape[0], x.shape[2] * x.shape[3]):
                return _PointwiseConvFn.apply(x, self.weight, _valid_wt(self, x.device))
        _stock("conv%dx%d%s" % (self.kernel_size[0], self.kernel_size[1], "_dw" if self.groups > 1 else ""), x)
        return F.conv2d(self.static_padding(x), self.weight, self.bias, self.stride, 0, self.dilation, self.groups)


def _pointwise_with_skip(conv, x):
    """(conv(x), x') for a block's expand conv whose input is also the block's skip connection: x' is x routed through the conv's autograd node
    (None when the libsrbh 1x1 path does not take this call: the caller keeps using x)"""
    if not (SKIP_THROUGH_EXPAND and conv._pointwise and x.is_cuda and x.dtype == torch.float32 and conv.weight.dtype == torch.float32 and x.dim() == 4
            and torch.is_grad_enabled() and x.requires_grad and PWCONV in ("1", "train")):
        return conv(x), None
    from . import _lib
    if not _lib.lib().srbh_pwconv_supported(x.shape[0], x.shape[1], conv.weight.shape[0], x.shape[2] * x.shape[3]):
        return conv(x), None
    return _PointwiseConvFn.apply(x, conv.weight, _valid_wt(conv, x.device), True)


def _valid_wt(conv, device):
    """the transposed copy PointwiseTransposes made of conv.weight, if it still belongs to this state of the weight (else None)"""
    wt = conv.__dict__.get("_srbh_wt")
    if wt is not None and (wt.device != device or conv.__dict__.get("_srbh_wt_state") != (
            conv.weight._version, conv.weight.data_ptr(), wcache.gen(conv.weight))):
        wt = None
    return wt


def _swish(x):
    # x * sigmoid(x) (efficientnet_pytorch's MemoryEfficientSwish) as ONE kernel forward and one backward: at 64x64 the
    # encoder is bound by the number of tiny launches, not by their work
    return F.silu(x)


def _drop_connect(x, p, training, mask=None):
    """stochastic depth (efficientnet_pytorch utils.drop_connect): x / keep * floor(keep + U[0,1)) per sample.  `mask` = this
    block's precomputed (B,) factor floor(keep + u) / keep (EfficientNetEncoder draws the uniforms of ALL blocks with one
    launch per forward instead of rand + add + floor + div + mul per block: 25 blocks x 5 launches)."""
    if not training or p <= 0:
        return x
    if mask is not None:
        return x * mask.view(-1, 1, 1, 1)
    keep = 1.0 - p
    mask = torch.floor(keep + torch.rand([x.shape[0], 1, 1, 1], dtype=x.dtype, device=x.device))
    return x / keep * mask


class MBConvBlock(nn.Module):
    def __init__(self, inp, out, kernel, stride, expand, image_size, se_ratio=0.25):
        super().__init__()
        self.inp, self.out, self.stride, self.expand = inp, out, stride, expand
        mid = inp * expand
        if expand != 1:
            self._expand_conv = SamePadConv2d(inp, mid, 1, image_size, bias=False)
            self._bn0 = nn.BatchNorm2d(mid, momentum=BN_MOM, eps=BN_EPS)
        self._depthwise_conv = SamePadConv2d(mid, mid, kernel, image_size, stride=stride, groups=mid, bias=False)
        self._bn1 = nn.BatchNorm2d(mid, momentum=BN_MOM, eps=BN_EPS)
        sq = max(1, int(inp * se_ratio))
        self._se_reduce = SamePadConv2d(mid, sq, 1, 1)
        self._se_expand = SamePadConv2d(sq, mid, 1, 1)
        self._project_conv = SamePadConv2d(mid, out, 1, math.ceil(image_size / stride), bias=False)
        self._bn2 = nn.BatchNorm2d(out, momentum=BN_MOM, eps=BN_EPS)

    def _eval_fused(self, x):
        """the whole block at inference on libsrbh's fused kernels, or None when a shape / mode is not theirs"""
        dw = self._depthwise_conv
        if not (MBCONV_EVAL and FUSED_SE_EVAL and PWCONV == "1" and x.dim() == 4 and _fused_eval_ok(self._bn1, x) and not self._bn2.training
                and (self.expand == 1 or not self._bn0.training) and dw._depthwise and self._project_conv._pointwise
                and dw.weight.dtype == torch.float32 and self._se_reduce.weight.is_contiguous() and self._se_expand.weight.is_contiguous()):
            return None
        from . import _lib
        L = _lib.lib()
        x = x.contiguous()
        B, _, H, W = x.shape
        mid, K, stride = dw.weight.shape[0], dw.weight.shape[-1], dw.stride[0]
        pl, pr, pt, pb = dw._pad
        OH, OW = (H + pt + pb - K) // stride + 1, (W + pl + pr - K) // stride + 1
        pc = self._project_conv
        out = pc.weight.shape[0]
        if not (L.srbh_dwconv_eval_supported(B, mid, H, W, K, stride, pt, pl, OH, OW) and L.srbh_pwconv_supported(B, mid, out, OH * OW)):
            return None
        st, dev = _lib.stream_ptr(), x.device
        if self.expand != 1:
            e = self._expand_conv(x).contiguous()
            a0, b0 = _bn_affine(self._bn0, dev)
        else:
            e, a0, b0 = x, None, None
        a1, b1 = _bn_affine(self._bn1, dev)
        y = torch.empty((B, mid, OH, OW), dtype=torch.float32, device=dev)
        pooled = torch.empty((B, mid), dtype=torch.float32, device=dev)
        _lib.check(L.srbh_dwconv_eval_fwd(e.data_ptr(), dw.weight.contiguous().data_ptr(), a0.data_ptr() if a0 is not None else None,
                                          b0.data_ptr() if b0 is not None else None, a1.data_ptr(), b1.data_ptr(), y.data_ptr(),
                                          pooled.data_ptr(), B, mid, H, W, K, stride, pt, pl, OH, OW, st), "dwconv_eval_fwd")
        SQ = self._se_reduce.out_channels
        hidden = torch.empty((B, SQ), dtype=torch.float32, device=dev)
        _lib.check(L.srbh_se_hidden(pooled.data_ptr(), self._se_reduce.weight.data_ptr(), self._se_reduce.bias.data_ptr(), hidden.data_ptr(),
                                    B, mid, SQ, st), "se_hidden")
        gate = torch.empty((B, mid), dtype=torch.float32, device=dev)
        _lib.check(L.srbh_se_gate(hidden.data_ptr(), self._se_expand.weight.data_ptr(), self._se_expand.bias.data_ptr(), gate.data_ptr(),
                                  B, mid, SQ, st), "se_gate")
        a2, b2 = _bn_affine(self._bn2, dev)
        wt = _valid_wt(pc, dev)
        w = wt if wt is not None else pc.weight.contiguous()
        z = torch.empty((B, out, OH, OW), dtype=torch.float32, device=dev)
        skip = self.stride == 1 and self.inp == self.out
        _lib.check(L.srbh_pwconv_fwd_epi(y.data_ptr(), w.data_ptr(), 1 if wt is not None else 0, z.data_ptr(), B, mid, out, OH * OW,
                                         gate.data_ptr(), a2.data_ptr(), b2.data_ptr(), x.data_ptr() if skip else None, 0, st), "pwconv_fwd_epi")
        return z

    def forward(self, x, drop_connect_rate=None, drop_mask=None):
        inputs = x
        if x.is_cuda and not self.training:
            z = self._eval_fused(x)
            if z is not None:
                return z
        if self.expand != 1 and x.is_cuda and self._bn0.training:
            if self.stride == 1 and self.inp == self.out:
                e_pre, through = _pointwise_with_skip(self._expand_conv, x)
                if through is not None:
                    inputs = through
            else:
                e_pre = self._expand_conv(x)
            MBm = _mbconv_train(self._bn0, e_pre)
            if (MBm is not None and MBm.se_supported(self._se_reduce, self._se_expand)
                    and MBm.mid_supported(self._bn0, self._bn1, self._depthwise_conv, e_pre)):
                # the block's middle as one libsrbh launch per direction (+ squeeze-excite's small kernels): csrc/srbh_mbconv.hip
                x = MBm.mid_se_train(self._bn0, e_pre, self._depthwise_conv, self._bn1, self._se_reduce, self._se_expand)
                return self._tail(x, inputs, drop_connect_rate, drop_mask)
            x = bn_act(self._bn0, e_pre, "silu")
        elif self.expand != 1:
            x = bn_act(self._bn0, self._expand_conv(x), "silu")
        x = self._depthwise_conv(x)
        MB = _mbconv_train(self._bn1, x)
        if MB is not None and MB.se_supported(self._se_reduce, self._se_expand):
            x = MB.bn_swish_se_train(self._bn1, x, self._se_reduce, self._se_expand)
        elif FUSED_SE_EVAL and _fused_eval_ok(self._bn1, x) and self._se_reduce.weight.is_contiguous() and self._se_expand.weight.is_contiguous():
            x = bn_swish_se(self._bn1, x, self._se_reduce, self._se_expand)
        else:
            x = bn_act(self._bn1, x, "silu")
            _stock("squeeze_excite", x)
            s = F.adaptive_avg_pool2d(x, 1)
            s = self._se_expand(_swish(self._se_reduce(s)))
            x = torch.sigmoid(s) * x
        return self._tail(x, inputs, drop_connect_rate, drop_mask)

    def _tail(self, x, inputs, drop_connect_rate, drop_mask):
        skip = self.stride == 1 and self.inp == self.out
        if skip and not (self.training and drop_connect_rate):
            return bn_act(self._bn2, self._project_conv(x), res=inputs)      # inference: BatchNorm + skip connection in one pass
        x = self._project_conv(x)
        if skip and drop_mask is not None:
            return bn_act(self._bn2, x, res=inputs, drop=drop_mask)           # training: BatchNorm, drop-connect and skip in one pass
        x = bn_act(self._bn2, x)
        if skip:
            if drop_connect_rate:
                x = _drop_connect(x, drop_connect_rate, self.training, drop_mask)
            x = x + inputs
        return x


def _mbconv_train(bn, x):
    """the libsrbh training kernels module if they take this BatchNorm input, else None"""
    if x.is_cuda and bn.training:
        from . import mbconv_autograd as MB
        if MB.supported(bn, x):
            return MB
    return None


class EfficientNetEncoder(nn.Module):
    """features = [x, stem, stage2, stage3, stage4, stage5] at strides (1,2,4,8,16,32); ``out_channels`` as in smp."""

    def __init__(self, name="efficientnet-b4", in_channels=3, depth=5):
        super().__init__()
        if name not in _EFFNET_PARAMS:
            raise NotImplementedError(f"encoder {name!r}: only the EfficientNet family used by the reference is restated "
                                      "(train.py:143 uses 'efficientnet-b4')")
        width, dmult, res, _ = _EFFNET_PARAMS[name]
        self._stage_idxs, chans = _SMP_STAGES[name]
        self._depth = depth
        self._in_channels = in_channels
        self.out_channels = tuple([in_channels] + list(chans[1:]))[:depth + 1]
        size = res
        stem = _round_filters(32, width)
        self._conv_stem = SamePadConv2d(in_channels, stem, 3, size, stride=2, bias=False)
        self._bn0 = nn.BatchNorm2d(stem, momentum=BN_MOM, eps=BN_EPS)
        size = math.ceil(size / 2)
        blocks = []
        for (rep, k, s, e, i, o) in _BASE_BLOCKS:
            i, o, rep = _round_filters(i, width), _round_filters(o, width), _round_repeats(rep, dmult)
            blocks.append(MBConvBlock(i, o, k, s, e, size))
            size = math.ceil(size / s)
            for _ in range(rep - 1):
                blocks.append(MBConvBlock(o, o, k, 1, e, size))
        self._blocks = nn.ModuleList(blocks)
        head = _round_filters(1280, width)
        self._conv_head = SamePadConv2d(blocks[-1].out, head, 1, size, bias=False)   # present but unused (as in smp)
        self._bn1 = nn.BatchNorm2d(head, momentum=BN_MOM, eps=BN_EPS)

    def forward(self, x):
        feats = [x]
        if x.is_cuda and x.dtype == torch.float32 and (PWCONV == "1" or (PWCONV == "train" and torch.is_grad_enabled())):
            pt = self.__dict__.get("_srbh_pwt")
            if pt is None:
                pt = self.__dict__["_srbh_pwt"] = PointwiseTransposes(
                    m for m in self.modules() if isinstance(m, SamePadConv2d) and m._pointwise and m is not self._conv_head)
            pt.refresh(x.device)
        x = _stem_eval(self._conv_stem, self._bn0, x) if _stem_eval_ok(self._conv_stem, self._bn0, x) else bn_act(self._bn0, self._conv_stem(x), "silu")
        feats.append(x)
        n = len(self._blocks)
        bounds = list(self._stage_idxs[:3]) + [n]
        masks = None
        if self.training and DROP_CONNECT > 0 and x.is_cuda:
            # all blocks' stochastic-depth factors from ONE uniform draw: floor(keep_i + u) / keep_i, keep_i = 1 - rate * i / n
            keep = (1.0 - DROP_CONNECT * torch.arange(n, dtype=x.dtype, device=x.device) / n).view(n, 1)
            masks = torch.floor(keep + torch.rand((n, x.shape[0]), dtype=x.dtype, device=x.device)) / keep     # [block][sample], rows contiguous
        for idx, blk in enumerate(self._blocks):
            if masks is not None:
                x = blk(x, DROP_CONNECT * idx / n, masks[idx])
                if idx + 1 in bounds:
                    feats.append(x)
                continue
            x = blk(x, DROP_CONNECT * idx / n)
            if idx + 1 in bounds:
                feats.append(x)
        return feats[:self._depth + 1]


def get_encoder(name, in_channels=3, depth=5, weights=None, **kwargs):
    """smp.encoders.get_encoder stand-in.  ``weights='imagenet'`` would download a checkpoint upstream; there is no
    network here, so the kwarg is accepted and the encoder stays randomly initialised unless SRBH_ENCODER_WEIGHTS
    points at a local efficientnet_pytorch state_dict (then smp's first-conv patch w[:, i] = w[:, i % 3] * 3/in_ch is
    applied for in_channels != 3)."""
    import os
    enc = EfficientNetEncoder(name, in_channels=in_channels, depth=depth)
    path = os.environ.get("SRBH_ENCODER_WEIGHTS")
    if weights is not None and path and os.path.isfile(path):
        sd = torch.load(path, map_location="cpu")
        sd = {k: v for k, v in sd.items() if not k.startswith("_fc.")}
        w = sd["_conv_stem.weight"]
        if in_channels != w.shape[1]:
            new = torch.empty(w.shape[0], in_channels, *w.shape[2:])
            for i in range(in_channels):
                new[:, i] = w[:, i % w.shape[1]]
            sd["_conv_stem.weight"] = new * (w.shape[1] / in_channels)
        enc.load_state_dict(sd, strict=True)
    return enc


# 3x3 convolutions of the U-Net decoders on libsrbh (csrc/srbh_dconv.hip) whenever the 16-bit operand policy of the head is active
# (hrfuse.head_h16(): the inference chain, TrainStep's 'f16' mode): fp16 forward, bf16 data / weight gradients, fp32 accumulation,
# NCHW fp32 tensors in and out.  The exact-fp32 modes keep the stock convolution (MIOpen), counted in STOCK_OPS.  SRBH_DCONV=0: never.
DCONV = __import__("os").environ.get("SRBH_DCONV", "1") == "1"


class _DecoderConvPacks:
    """fragment-order 16-bit images of one decoder conv weight: fp16 for the forward, bf16 transposed + flipped for the data gradient;
    remade when the weight changes (version / address / generation stamp: fused optimizers and graph replays do not bump _version)"""

    def __init__(self):
        self.kf = self.kb = None
        self.f = self.b = None

    @staticmethod
    def _key(w):
        return (w._version, w.data_ptr(), wcache.gen(w), str(w.device))

    def fwd(self, w):
        from . import _lib
        k = self._key(w)
        if k != self.kf:
            L = _lib.lib()
            cout, cin = w.shape[:2]
            wc = w.detach().float().contiguous()
            self.f = torch.empty(L.srbh_hpack_h16_bytes(cout, cin, 3) // 2, dtype=torch.int16, device=w.device)
            _lib.check(L.srbh_hpack_conv_h16(wc.data_ptr(), cout, cin, 3, 0, 0, self.f.data_ptr(), _lib.stream_ptr()), "hpack_conv_h16")
            self.kf = k
        wcache.keep(self.f)
        return self.f

    def bwd(self, w):
        from . import _lib
        k = self._key(w)
        if k != self.kb:
            L = _lib.lib()
            cout, cin = w.shape[:2]
            wc = w.detach().float().contiguous()
            self.b = torch.empty(L.srbh_hpack_h16_bytes(cin, cout, 3) // 2, dtype=torch.int16, device=w.device)
            _lib.check(L.srbh_hpack_conv_h16(wc.data_ptr(), cin, cout, 3, 1, 1, self.b.data_ptr(), _lib.stream_ptr()), "hpack_conv_h16")
            self.kb = k
        wcache.keep(self.b)
        return self.b


class DecoderPackTable:
    """Both 16-bit images of every 3x3 conv of a decoder in ONE flat buffer, refreshed by ONE launch (srbh_dconv_pack_many) at the start
    of a forward whenever a weight changed -- in training that is once per step (per-conv packing was 2 launches per conv and step:
    80 launches, 0.44 ms, profiles/r04m_train_steady_kernel_stats.txt).  Fills the convs' _DecoderConvPacks in place."""

    def __init__(self, convs):
        self.convs = list(convs)
        self.key = None
        self.flat = self.table = None
        self.ptrs = None

    def refresh(self, device):
        from . import _lib
        import numpy as np
        key = (tuple(_DecoderConvPacks._key(c.weight) for c in self.convs), str(device))
        if key == self.key:
            wcache.keep(self.flat, self.table)
            return
        L = _lib.lib()
        ptrs = tuple(c.weight.data_ptr() for c in self.convs)
        if self.flat is None or self.flat.device != device or ptrs != self.ptrs:
            sizes = [L.srbh_hpack_h16_bytes(c.weight.shape[0], c.weight.shape[1], 3) // 2 for c in self.convs]
            self.flat = torch.empty(2 * sum(sizes), dtype=torch.int16, device=device)
            desc = np.zeros(len(self.convs), dtype=np.dtype([("w", "<u8"), ("fwd", "<u8"), ("bwd", "<u8"), ("cout", "<i4"), ("cin", "<i4")]))
            o = 0
            self.views = []
            for i, (c, n) in enumerate(zip(self.convs, sizes)):
                f, b = self.flat[o:o + n], self.flat[o + n:o + 2 * n]
                desc[i] = (c.weight.data_ptr(), f.data_ptr(), b.data_ptr(), c.weight.shape[0], c.weight.shape[1])
                self.views.append((f, b))
                o += 2 * n
            self.table = torch.from_numpy(desc.view(np.uint8).copy()).to(device)
            self.ptrs = ptrs
        _lib.check(L.srbh_dconv_pack_many(self.table.data_ptr(), len(self.convs), _lib.stream_ptr()), "dconv_pack_many")
        for c, (f, b), k in zip(self.convs, self.views, key[0]):
            pk = c.__dict__.get("_srbh_dconv_packs")
            if pk is None:
                pk = c.__dict__["_srbh_dconv_packs"] = _DecoderConvPacks()
            pk.f, pk.b, pk.kf, pk.kb = f, b, k, k
        self.key = key
        wcache.keep(self.flat, self.table)


class _DecoderConvFn(torch.autograd.Function):
    """y = conv3x3(x, w) (stride 1, padding 1, no bias) on libsrbh: one launch forward, one for the input gradient, two (partials +
    ordered reduce) for the weight gradient -- MIOpen: fp32 Winograd + NHWC implicit-GEMM weight gradients behind batched transposes
    and zero fills, ~6 launches per conv and step."""

    @staticmethod
    def forward(ctx, x, weight, packs):
        from . import _lib
        x = x.contiguous()
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().srbh_dconv_fwd(x.data_ptr(), packs.fwd(weight).data_ptr(), y.data_ptr(), B, Cin, Cout, H, W, 0,
                                             _lib.stream_ptr()), "dconv_fwd")
        ctx.save_for_backward(x, weight)
        ctx.packs = packs
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        x, weight = ctx.saved_tensors
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dy = dy.contiguous()
        L = _lib.lib()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.srbh_dconv_fwd(dy.data_ptr(), ctx.packs.bwd(weight).data_ptr(), dx.data_ptr(), B, Cout, Cin, H, W, 1,
                                        _lib.stream_ptr()), "dconv_fwd (data gradient)")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            ws = torch.empty(L.srbh_dconv_wgrad_ws_floats(B, Cin, Cout, H, W), dtype=torch.float32, device=x.device)
            _lib.check(L.srbh_dconv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, Cin, Cout, H, W, _lib.stream_ptr()),
                       "dconv_wgrad")
        return dx, dw, None


def decoder_conv(conv, x):
    """conv(x) for a decoder 3x3 conv: libsrbh when the 16-bit operand policy is active and the shape is taken, else the module"""
    if (DCONV and x.is_cuda and x.dtype == torch.float32 and conv.bias is None and conv.weight.dtype == torch.float32 and x.dim() == 4
            and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.groups == 1):
        from . import _lib
        from . import hrfuse as _H
        if _H.head_h16() and _lib.lib().srbh_dconv_supported(x.shape[0], x.shape[1], conv.weight.shape[0], x.shape[2], x.shape[3]):
            packs = conv.__dict__.get("_srbh_dconv_packs")
            if packs is None:
                packs = conv.__dict__["_srbh_dconv_packs"] = _DecoderConvPacks()
            return _DecoderConvFn.apply(x, conv.weight, packs)
    _stock("decoder_conv3x3", x)             # (MIOpen: the exact-fp32 modes, CPU tensors are not counted)
    return conv(x)


DCONV_EVAL_EPI = __import__("os").environ.get("SRBH_DCONV_EVAL_EPI", "1") == "1"      # inference: BatchNorm + ReLU in the decoder conv's store


def _decoder_conv_bn_relu_eval(conv, bn, x):
    """inference: relu(bn(conv(x))) of a decoder block as ONE libsrbh launch (srbh_dconv_fwd_epi), or None when not applicable"""
    if not (DCONV and DCONV_EVAL_EPI and _fused_eval_ok(bn, x) and x.dim() == 4 and conv.bias is None and conv.weight.dtype == torch.float32
            and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.groups == 1):
        return None
    from . import _lib
    from . import hrfuse as _H
    B, Cin, H, W = x.shape
    Cout = conv.weight.shape[0]
    if not (_H.head_h16() and _lib.lib().srbh_dconv_supported(B, Cin, Cout, H, W)):
        return None
    packs = conv.__dict__.get("_srbh_dconv_packs")
    if packs is None:
        packs = conv.__dict__["_srbh_dconv_packs"] = _DecoderConvPacks()
    scale, shift = _bn_affine(bn, x.device)
    x = x.contiguous()
    y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().srbh_dconv_fwd_epi(x.data_ptr(), packs.fwd(conv.weight).data_ptr(), y.data_ptr(), B, Cin, Cout, H, W, 0, scale.data_ptr(),
                                             shift.data_ptr(), 2, _lib.stream_ptr()), "dconv_fwd_epi")
    return y


class _ConvBnRelu(nn.Sequential):
    def __init__(self, cin, cout, use_batchnorm=True):
        mods = [nn.Conv2d(cin, cout, 3, padding=1, bias=not use_batchnorm)]
        if use_batchnorm:
            mods.append(nn.BatchNorm2d(cout))
        mods.append(nn.ReLU(inplace=True))
        super().__init__(*mods)
        self._bn = use_batchnorm

    def forward(self, x):
        if self._bn:
            if not self[1].training and x.is_cuda:
                y = _decoder_conv_bn_relu_eval(self[0], self[1], x)
                if y is not None:
                    return y
            return bn_act(self[1], decoder_conv(self[0], x), "relu")
        return super().forward(x)


class _Attention(nn.Module):
    def __init__(self, name):
        super().__init__()
        if name is not None:
            raise NotImplementedError("only attention_type=None is used by the reference (mymodels.py:251,258)")
        self.attention = nn.Identity()

    def forward(self, x):
        return self.attention(x)


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, skip_channels, out_channels, use_batchnorm=True, attention_type=None):
        super().__init__()
        self.conv1 = _ConvBnRelu(in_channels + skip_channels, out_channels, use_batchnorm)
        self.attention1 = _Attention(attention_type)
        self.conv2 = _ConvBnRelu(out_channels, out_channels, use_batchnorm)
        self.attention2 = _Attention(attention_type)

    def forward(self, x, skip=None):
        if x.is_cuda:
            from . import mbconv_autograd as MB
            if MB.up2_cat_supported(x, skip):                    # nearest x2 + concat as ONE libsrbh launch (and one backward)
                x = MB.up2_cat(x, skip)
                if skip is not None:
                    x = self.attention1(x)
                return self.attention2(self.conv2(self.conv1(x)))
        _stock("up2_cat", x)
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if skip is not None:
            x = self.attention1(torch.cat([x, skip], dim=1))
        return self.attention2(self.conv2(self.conv1(x)))


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels, n_blocks=5, use_batchnorm=True, attention_type=None,
                 center=False):
        super().__init__()
        if n_blocks != len(decoder_channels):
            raise ValueError(f"Model depth is {n_blocks}, but you provide `decoder_channels` for "
                             f"{len(decoder_channels)} blocks.")
        if center:
            raise NotImplementedError("center block is only used with VGG encoders (mymodels.py:250)")
        enc = list(encoder_channels[1:])[::-1]
        ins = [enc[0]] + list(decoder_channels[:-1])
        skips = enc[1:] + [0]
        self.center = nn.Identity()
        self.blocks = nn.ModuleList(DecoderBlock(i, s, o, use_batchnorm, attention_type)
                                    for i, s, o in zip(ins, skips, decoder_channels))

    def forward(self, *features):
        feats = features[1:][::-1]
        x = self.center(feats[0])
        if DCONV and x.is_cuda and x.dtype == torch.float32:
            from . import hrfuse as _H
            if _H.head_h16():            # the decoder convs will run on libsrbh: one pack launch for all ten (when a weight changed)
                pt = self.__dict__.get("_srbh_dpt")
                if pt is None:
                    pt = self.__dict__["_srbh_dpt"] = DecoderPackTable(
                        b_[0] for blk in self.blocks for b_ in (blk.conv1, blk.conv2) if isinstance(b_[0], nn.Conv2d) and b_[0].bias is None)
                if pt.convs and all(c.weight.dtype == torch.float32 for c in pt.convs):
                    pt.refresh(x.device)
        skips = feats[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x
