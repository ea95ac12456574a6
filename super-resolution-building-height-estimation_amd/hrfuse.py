"""MI355X-native HR feature / fusion / regression head behind the reference's SR/HRfuse.py interface.

Mirrors (constructor kwargs, forward signatures, state_dict keys) of the reference classes
``Upsampler`` (SR/HRfuse.py:17-44), ``BasicBlock`` (:109-159), ``HRfeature`` (:164-169),
``HRfuse_residual`` (:173-190) and the sibling variants ``HRfuse`` (:47-65), ``HRfuse_x2`` (:68-90),
``HRupsample`` (:193-202), ``GeoNet`` (:205-214), ``Refine_residual`` (:217-228).

nn.Conv2d / nn.BatchNorm2d sub-modules are parameter containers only; every forward runs libsrbh's
fp32 matrix-core kernels (csrc/srbh_head.hip) on NHWC (channels_last) tensors:
  conv (+concat, +producer's BN+ReLU folded into the load, +PixelShuffle folded into the store,
  +BatchNorm batch statistics in the epilogue) -> bn finalize -> fused bn+add+relu.
Training-mode forward/backward is provided by the autograd functions in ``hrfuse_autograd``.
No CPU / eager fallback: a non-ROCm input raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Optional

import torch
from torch import nn

from . import _lib, wcache

__all__ = ["default_conv", "Upsampler", "conv3x3", "conv1x1", "BasicBlock", "HRfeature", "HRfuse_residual",
           "HRfuse", "HRfuse_x2", "HRupsample", "GeoNet", "Refine_residual"]

BN_EPS_DEFAULT = 1e-5


# ----------------------------------------------------------------------------- low-level ops (NHWC fp32)
_T16 = (torch.float16, torch.bfloat16)      # 16-bit tensors in memory: fp16 activations (fp16-operand kernels), bf16 gradients (bf16-operand kernels)


def _require_dev(x, who, h16_ok=False):
    if not (torch.is_tensor(x) and x.is_cuda):
        raise RuntimeError(f"{who} (libsrbh): input must be a ROCm/HIP device tensor; the head has no CPU fallback")
    if x.dtype != torch.float32 and not (h16_ok and x.dtype == torch.float16):
        raise TypeError(f"{who}: expected float32, got {x.dtype}")


def to_nhwc(x):
    """(B,C,H,W) tensor -> same logical tensor whose memory is [B][H][W][C] (no copy if it already is)."""
    if x.dim() != 4:
        raise ValueError(f"expected a (B,C,H,W) tensor, got {tuple(x.shape)}")
    B, Cc, H, W = x.shape
    if x.stride() == (H * W * Cc, 1, W * Cc, Cc):
        return x
    if Cc == 1 and x.is_contiguous():
        return x.as_strided((B, 1, H, W), (H * W, 1, W, 1))
    src = x.contiguous()
    out = torch.empty_strided((B, Cc, H, W), (H * W * Cc, 1, W * Cc, Cc), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().srbh_nchw_to_nhwc_f32(src.data_ptr(), out.data_ptr(), B, Cc, H, W, _lib.stream_ptr()),
               "nchw_to_nhwc")
    return out


def empty_nhwc(B, Cc, H, W, device, dtype=torch.float32):
    return torch.empty_strided((B, Cc, H, W), (H * W * Cc, 1, W * Cc, Cc), dtype=dtype, device=device)


# ---- operand precision of the head convolutions ---------------------------------------------------------------------
# "f32": exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32), <= 2e-5 against the reference -- the strict mode;
# "f16": the staged activations and the weights are rounded to fp16, fp32 accumulate (srbh_hconv_h16): ~1/8 of the
#        matrix-core time, the kernel then runs at its HBM traffic (BASELINE configs[4]: "fp16 MFMA");
# "auto" (default): f16 under torch.no_grad() / eval (inference, the predict path), f32 when a graph is recorded
#        (training: forward, data- and weight-gradients stay exact).  Override: SRBH_HEAD_PRECISION or set_head_precision().
import os as _os

_HEAD_PRECISION = {"mode": _os.environ.get("SRBH_HEAD_PRECISION", "auto"), "depth": 0}   # depth > 0: inside an autograd Function


def set_head_precision(mode="auto"):
    if mode not in ("auto", "f16", "f32"):
        raise ValueError("head precision must be 'auto', 'f16' or 'f32'")
    _HEAD_PRECISION["mode"] = mode


class head_precision:
    """scoped `set_head_precision`: `with head_precision("f16"): ...` restores the previous mode on exit (harness.TrainStep uses
    it around its step so that constructing a TrainStep does not change what the rest of the process computes)."""

    def __init__(self, mode):
        if mode not in ("auto", "f16", "f32"):
            raise ValueError("head precision must be 'auto', 'f16' or 'f32'")
        self.mode = mode

    def __enter__(self):
        self.prev = _HEAD_PRECISION["mode"]
        _HEAD_PRECISION["mode"] = self.mode
        return self

    def __exit__(self, *exc):
        _HEAD_PRECISION["mode"] = self.prev
        return False


def head_h16():
    m = _HEAD_PRECISION["mode"]
    # (torch disables grad mode inside autograd.Function.forward / backward: those bump "depth" instead, hrfuse_autograd._exact)
    return m == "f16" or (m == "auto" and not torch.is_grad_enabled() and _HEAD_PRECISION["depth"] == 0)


# fp16 ACTIVATIONS in memory between the convs of an inference chain (BasicBlock.forward_nhwc; SRBH_FP16_ACT=0 switches it off).  ON by
# default since round 3: it halves the bytes every head conv moves (32 + 32 instead of 64 + 64 per pixel; height maps 4e-4 from the
# fp32-tensor chain, inside the 1e-3 tolerance).  History: with the round-2 one-tile-per-workgroup kernels it measured 3 % SLOWER
# (model forward 15.0 vs 14.55 ms at B=128); on the persistent double-buffered hconv16 forms of round 3 it is +3.7 % tiles/s on the
# tiled predict path, hence the default.  Covered by tests/test_gpu_head_f16.py in both settings.
FP16_ACTIVATIONS = _os.environ.get("SRBH_FP16_ACT", "1") == "1"

# 16-bit tensors INSIDE a BasicBlock of the training step (hrfuse_autograd._BasicBlockFn, mixed-precision mode "f16" only; round 3).
# (1) TRAIN_IO16: the gradient tensors that never leave one block's backward -- dz, dc2, da1, dc1, dd -- are stored as bf16.  Their
#     consumers, the data- and weight-gradient kernels, round their operands to bf16 anyway, and the BatchNorm-backward passes read
#     them widened to fp32: measured against the exact-fp32 graph the parameter gradients are as accurate as with fp32 tensors
#     (median relative error 7.4e-3 both ways, tools/io16_err.py) -- 14 of a plain block's 21 backward tensor passes move half the
#     bytes, the step 49.3 -> 47.0 ms.  SRBH_TRAIN_IO16=0 keeps every tensor fp32 (A/B aid).
# (2) TRAIN_IO16_ACT: additionally store saved ACTIVATIONS as fp16: "c1c2" = conv1's, conv2's and the downsample conv's outputs,
#     "c1" / "c2" = one of them, "none" (default) = none.  "c1c2" is another 2 ms (45.0 ms) but costs accuracy the default mode does
#     not spend: the consumer rounds an already-rounded value (training-mode forward of HRfeature + HRfuse_residual against the exact
#     graph: 8.9e-4 -> 1.11e-3, i.e. past the 1e-3 the mixed mode otherwise keeps; gradients unchanged at cos 0.996).  Opt in.
# Block inputs / outputs and the gradients crossing autograd stay fp32 either way.
TRAIN_IO16 = _os.environ.get("SRBH_TRAIN_IO16", "1") == "1"
TRAIN_IO16_ACT = _os.environ.get("SRBH_TRAIN_IO16_ACT", "none")


def fp16_chain(mod):
    """True when `mod`'s blocks run the fp16-activation inference chain right now (eval mode, no graph, fp16-operand precision)"""
    return FP16_ACTIVATIONS and not mod.training and not torch.is_grad_enabled() and head_h16()


# ---- the 16-bit packs of TRAINED weights, refreshed right behind the optimizer (round 5) ----------------------------------------------
# In the training step every head conv weight is packed twice per step (forward: fp16, data gradient: bf16 transposed + flipped) because
# the optimizer changed it: 56 launches of ~5 us, each in front of a chip-filling kernel.  A pack of a bias-free conv whose weight is a
# Parameter registers itself here when it is made; wcache's optimizer post-step hook calls `repack_after_step`, which rewrites all registered
# packs of the stepped parameters IN PLACE with one launch (srbh_hpack_conv_h16_many) and moves their cache keys to the new generation, so
# that the next `.get()` finds them current.  A pack that is not registered (bias, fp32 form, a weight view) keeps repacking lazily.
# SRBH_PACK_AFTER_STEP=0: off.
PACK_AFTER_STEP = _os.environ.get("SRBH_PACK_AFTER_STEP", "1") == "1"


class _PackRegistry:
    def __init__(self):
        self.entries = {}          # id(cache) -> dict(cache=weakref, param=weakref, buf, args, rekey)
        self.tables = {}           # tuple(entry ids) -> (device table, max elements, keep-alive)

    def register(self, cache, param, buf, args, rekey):
        import weakref
        if not (PACK_AFTER_STEP and isinstance(param, nn.Parameter) and param.dtype == torch.float32 and param.is_contiguous() and param.is_cuda):
            return
        self.entries[id(cache)] = dict(cache=weakref.ref(cache), param=weakref.ref(param), buf=buf, args=args, rekey=rekey,
                                       ptr=param.data_ptr(), dev=param.device)
        self.tables.clear()

    def repack(self, params):
        if not self.entries:
            return
        import numpy as np
        ids = {id(p) for p in params}
        live, dead = [], []
        for k, e in self.entries.items():
            c, p = e["cache"](), e["param"]()
            if c is None or p is None or c.w is not e["buf"] or p.data_ptr() != e["ptr"]:
                dead.append(k)                       # the module is gone, or the pack / the parameter storage was replaced (the lazy path re-registers)
            elif id(p) in ids and not torch.cuda.is_current_stream_capturing():
                live.append(k)
        for k in dead:
            del self.entries[k]
        if dead:
            self.tables.clear()
        by_dev = {}
        for k in live:
            by_dev.setdefault(self.entries[k]["dev"], []).append(k)
        L = _lib.lib()
        for dev, keys in by_dev.items():
            tk = tuple(keys)
            t = self.tables.get(tk)
            if t is None:
                desc = np.zeros(len(keys), dtype=np.dtype([("w", "<u8"), ("out", "<u8"), ("cout", "<i4"), ("cin", "<i4"), ("ks", "<i4"), ("tf", "<i4"),
                                                           ("bf", "<i4"), ("pad", "<i4")]))
                mx = 0
                for i, k in enumerate(keys):
                    e = self.entries[k]
                    co, ci, ks, tf, bf = e["args"]
                    desc[i] = (e["ptr"], e["buf"].data_ptr(), co, ci, ks, tf, bf, 0)
                    mx = max(mx, L.srbh_hpack_h16_bytes(co, ci, ks) // 2)
                t = self.tables[tk] = (torch.from_numpy(desc.view(np.uint8).copy()).to(dev), mx)
            with torch.cuda.device(dev):
                _lib.check(L.srbh_hpack_conv_h16_many(t[0].data_ptr(), len(keys), t[1], _lib.stream_ptr()), "hpack_conv_h16_many")
            for k in keys:
                e = self.entries[k]
                c = e["cache"]()
                nk = e["rekey"]()
                if c is not None and nk is not None:
                    c.key = nk


PACKS = _PackRegistry()


def repack_after_step(optimizer):
    """called by wcache's optimizer post-step hook, behind the generation stamp"""
    if PACK_AFTER_STEP and PACKS.entries:
        PACKS.repack([p for g in optimizer.param_groups for p in g["params"]])


class _PackedConv:
    """HWPACK32 (or its fp16 form) of one conv weight (+ zero-padded bias), rebuilt when the parameter changes."""

    def __init__(self):
        self.key = None
        self.w = None
        self.b = None
        self._other = None        # the pack in the OTHER row order (up / not up): calls that alternate between the two forms do not repack

    def get(self, conv: nn.Conv2d, h16=False, up=False):
        """up: the Upsampler kernel's SUB-PIXEL-MAJOR row order (include/srbh.h, srbh_hconv_args.pixelshuffle2 == 2): packed row
        ob*16 + kk*4 + q = conv channel (kk*4 + ob)*4 + q of a 64-channel weight (and bias)"""
        w = conv.weight
        key = (w._version, w.data_ptr(), None if conv.bias is None else (conv.bias._version, conv.bias.data_ptr()), bool(h16), bool(up),
               wcache.gen(w, conv.bias))            # (fused optimizers do not bump _version, see wcache.py)
        if key != self.key and self._other is not None and self._other[0] == key:       # one slot per pack order
            self._other, (self.key, self.w, self.b) = (self.key, self.w, self.b), self._other
        if key != self.key:
            if self.key is not None and self.key[4] != key[4]:
                self._other = (self.key, self.w, self.b)
            L = _lib.lib()
            cout, cin, ks, _ = w.shape
            wc = w.detach().float().contiguous()
            perm = None
            if up:
                perm = _up_perm(w.device)
                wc = wc.index_select(0, perm)
            if h16:
                buf = torch.empty(L.srbh_hpack_h16_bytes(cout, cin, ks) // 2, dtype=torch.float16, device=w.device)
                _lib.check(L.srbh_hpack_conv_h16(wc.data_ptr(), cout, cin, ks, 0, 0, buf.data_ptr(), _lib.stream_ptr()), "hpack_conv_h16")
            else:
                buf = torch.empty(L.srbh_hpack_bytes(cout, cin, ks) // 4, dtype=torch.float32, device=w.device)
                _lib.check(L.srbh_hpack_conv_f32(wc.data_ptr(), cout, cin, ks, 0, buf.data_ptr(), _lib.stream_ptr()),
                           "hpack_conv_f32")
            b = None
            if conv.bias is not None:
                b = torch.zeros((cout + 15) // 16 * 16, dtype=torch.float32, device=w.device)
                b[:cout] = conv.bias.detach().float() if perm is None else conv.bias.detach().float().index_select(0, perm)
            # (no host sync: `wc` is recycled by torch's stream-ordered allocator, and the pack kernel runs on that stream)
            self.key, self.w, self.b = key, buf, b
            if h16 and not up and conv.bias is None:           # refreshed behind the optimizer's step from now on (PACKS)
                import weakref
                cref = weakref.ref(conv)

                def rekey(cref=cref):
                    cv = cref()
                    if cv is None or cv.bias is not None:
                        return None
                    ww = cv.weight
                    return (ww._version, ww.data_ptr(), None, True, False, wcache.gen(ww, None))
                PACKS.register(self, w, buf, (cout, cin, ks, 0, 0), rekey)
        wcache.keep(self.w, self.b)
        return self.w, self.b


# the Upsampler's 16 -> 64 conv + PixelShuffle(2) on its own persistent kernel (csrc/srbh_hconv_up_kernel.h) whenever the fp16-operand forms
# run and the shape is the kernel's; SRBH_HCONV_UP=0: the template with its LDS-ordered PixelShuffle store (A/B aid; same bits)
HCONV_UP = _os.environ.get("SRBH_HCONV_UP", "1") == "1"
_UP_PERM = {}


def _up_perm(device):
    """perm[ob*16 + kk*4 + q] = (kk*4 + ob)*4 + q: which conv channel sits in which row of the sub-pixel-major pack"""
    key = str(device)
    t = _UP_PERM.get(key)
    if t is None:
        t = _UP_PERM[key] = torch.tensor([(kk * 4 + ob) * 4 + q for ob in range(4) for kk in range(4) for q in range(4)], dtype=torch.int64,
                                         device=device)
    return t


# ---- BatchNorm partial-sum buffers: a pool of ZEROED buffers (round 5) -------------------------------------------------------------
# Every statistics producer (conv epilogues, the BatchNorm-backward reduce pass) used to zero its buffer with a launch of its own:
# 49 one-line kernels per training step, each a dependent launch in front of a convolution.  Now the consumer cleans up: the
# finalize kernels zero the slots behind their read (srbh_bn_finalize_clear / srbh_bn_bwd_finalize_clear) and the buffer goes back
# to a per-(size, device, stream) free list; a producer that takes it from there passes stats_clean = 1 and launches no fill.
# Outside the pool (stream capture: the buffers would belong to the graph's private pool; SRBH_STATS_POOL=0) nothing changes.
STATS_POOL = _os.environ.get("SRBH_STATS_POOL", "1") == "1"
_STATS_FREE = {}


def stats_acquire(C16, device):
    """-> a float64 partial-sum buffer for C16 channels; `buf._srbh_clean` tells the producer whether it is known to be all zero"""
    n = _lib.lib().srbh_bn_stats_bytes(C16) // 8
    dev = torch.device(device)
    if not STATS_POOL or dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
        buf = torch.empty(n, dtype=torch.float64, device=dev)
        buf._srbh_clean = False
        return buf
    key = (n, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    free = _STATS_FREE.setdefault(key, [])
    buf = free.pop() if free else torch.zeros(n, dtype=torch.float64, device=dev)
    buf._srbh_clean = True
    buf._srbh_key = key
    return buf


def stats_clean(buf):
    return bool(getattr(buf, "_srbh_clean", False))


def stats_release(buf):
    """call right after a *_finalize_clear on `buf` was queued (same stream): the buffer is zero again for whoever takes it next"""
    key = getattr(buf, "_srbh_key", None)
    if key is not None and stats_clean(buf) and len(_STATS_FREE.setdefault(key, [])) < 64:
        _STATS_FREE[key].append(buf)


def _hconv_args(srcs, conv: nn.Conv2d, packed: _PackedConv, pre=None, ps2=False, want_stats=False, post=None, res=None,
                post_relu=False, out_h16=False):
    """conv(cat(srcs)) through srbh_hconv_f32.  srcs: list of 1..2 NHWC tensors; pre=(scale, shift, relu) is
    applied to srcs[0]; returns (out, stats) with out NHWC and stats the partial-sum buffer or None.
    Epilogue extras (inference fusion): post=(scale, shift) per output channel, res = NHWC tensor added, post_relu.
    fp16 activations in memory (inference chain, fp16-operand mode only): any of srcs / res may be a torch.float16 NHWC tensor,
    out_h16=True writes one (the epilogue rounds once; the consumer stages the halves as they are)."""
    L = _lib.lib()
    x0 = srcs[0]
    B, c0, H, W = x0.shape
    c1 = srcs[1].shape[1] if len(srcs) > 1 else 0
    cout, cin, ks, _ = conv.weight.shape
    if cin != c0 + c1:
        raise ValueError(f"conv expects {cin} input channels, got {c0}+{c1}")
    h16 = head_h16()
    up = bool(ps2 and h16 and HCONV_UP and pre is None and post is None and res is None and not want_stats and not post_relu and cout == 64
              and c0 == 16 and c1 == 0 and ks == 3 and L.srbh_hconv_up_supported(H, W))
    w, b = packed.get(conv, h16, up)
    a = _lib.HConvArgs()
    a.src0, a.c0 = x0.data_ptr(), c0
    if pre is not None:
        a.pre_scale, a.pre_shift, a.pre_relu = pre[0].data_ptr(), pre[1].data_ptr(), int(pre[2])
    if c1:
        a.src1, a.c1 = srcs[1].data_ptr(), c1
    a.w = w.data_ptr()
    a.bias = b.data_ptr() if b is not None else None
    a.cout, a.ksize = cout, ks
    a.B, a.H, a.W = B, H, W
    a.pixelshuffle2 = 2 if up else int(ps2)
    io = ((1 if x0.dtype in _T16 else 0) | (2 if c1 and srcs[1].dtype in _T16 else 0)
          | (4 if res is not None and res.dtype in _T16 else 0) | (8 if out_h16 else 0))
    if io and not h16:
        raise RuntimeError("libsrbh hconv: fp16 activations need the fp16-operand mode (set_head_precision)")
    if any(t is not None and t.dtype == torch.bfloat16 for t in (x0, srcs[1] if c1 else None, res)):
        raise TypeError("libsrbh hconv (forward, fp16 operands): 16-bit tensors must be float16 (bfloat16 is the gradient kernels' type)")
    a.io_h16 = io
    odt = torch.float16 if out_h16 else torch.float32
    out = empty_nhwc(B, cout // 4, 2 * H, 2 * W, x0.device, odt) if ps2 else empty_nhwc(B, cout, H, W, x0.device, odt)
    a.out = out.data_ptr()
    if post is not None:
        a.post_scale, a.post_shift = post[0].data_ptr(), post[1].data_ptr()
    if res is not None:
        a.res1, a.res1_ld, a.res1_scale = res.data_ptr(), res.shape[1], 1.0
    a.post_relu = int(post_relu)
    stats = None
    if want_stats:
        stats = stats_acquire((cout + 15) // 16 * 16, x0.device)
        a.stats, a.stats_clean = stats.data_ptr(), int(stats_clean(stats))
    return a, out, stats, h16, (w, b)          # (w, b: keep the packed buffers alive until the launch is issued)


def hconv(srcs, conv: nn.Conv2d, packed: _PackedConv, pre=None, ps2=False, want_stats=False, post=None, res=None,
          post_relu=False, out_h16=False):
    """conv(cat(srcs)) through srbh_hconv_f32 / srbh_hconv_h16: see _hconv_args for the arguments; returns (out, stats)."""
    L = _lib.lib()
    a, out, stats, h16, _keep = _hconv_args(srcs, conv, packed, pre, ps2, want_stats, post, res, post_relu, out_h16)
    if h16:
        _lib.check(L.srbh_hconv_h16(C.byref(a), 0, _lib.stream_ptr()), "hconv_h16")
    else:
        _lib.check(L.srbh_hconv_f32(C.byref(a), _lib.stream_ptr()), "hconv_f32")
    return out, stats


# a plain BasicBlock of the fp16 inference chain as one libsrbh pass (csrc/srbh_hblock16_kernel.h); SRBH_HBLOCK16=0: conv1 and conv2 as two launches (A/B aid)
HBLOCK16 = _os.environ.get("SRBH_HBLOCK16", "1") == "1"


def hblock16(x, blk, bn1, bn2, out_h16):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + x) for a 16-channel fp16 NHWC `x`, folded BatchNorms bn1 / bn2 = (scale, shift)"""
    L = _lib.lib()
    B, _, H, W = x.shape
    w1, _ = blk._p1.get(blk.conv1, True, False)
    w2, _ = blk._p2.get(blk.conv2, True, False)
    a = _lib.HBlock16Args()
    a.x, a.w1, a.w2 = x.data_ptr(), w1.data_ptr(), w2.data_ptr()
    a.scale1, a.shift1, a.scale2, a.shift2 = bn1[0].data_ptr(), bn1[1].data_ptr(), bn2[0].data_ptr(), bn2[1].data_ptr()
    out = empty_nhwc(B, 16, H, W, x.device, torch.float16 if out_h16 else torch.float32)
    a.out, a.out_h16 = out.data_ptr(), int(out_h16)
    a.B, a.H, a.W = B, H, W
    _lib.check(L.srbh_hblock16_eval(C.byref(a), _lib.stream_ptr()), "hblock16_eval")
    return out


def hconv_entry(srcs, conv1, packed1, convd, packedd, want_stats=False, postd=None, post1=None, post1_relu=False, out_h16=False):
    """The entry of a BasicBlock with a downsample branch: conv1 (3x3) and downsample[0] (1x1) over the same cat(srcs) in ONE
    libsrbh call (srbh_hconv_entry_h16: one fused pass over the input when the shapes allow, else the two launches; 16-bit operand
    modes only).  Returns (c1, stats1, d, statsd); postd = (scale, shift) of the downsample BatchNorm in inference; post1 /
    post1_relu = bn1 (+ ReLU) in conv1's epilogue and out_h16 = both outputs as fp16 tensors (the fp16 inference chain, and the
    fp16 saved activations of the training step)."""
    L = _lib.lib()
    a1, c1, st1, h16, _k1 = _hconv_args(srcs, conv1, packed1, want_stats=want_stats, post=post1, post_relu=post1_relu, out_h16=out_h16)
    a2, d, st2, _, _k2 = _hconv_args(srcs, convd, packedd, want_stats=want_stats, post=postd, out_h16=out_h16)
    if not h16:
        raise RuntimeError("hconv_entry: fp16-operand mode only")
    _lib.check(L.srbh_hconv_entry_h16(C.byref(a1), C.byref(a2), 0, _lib.stream_ptr()), "hconv_entry_h16")
    return c1, st1, d, st2


# ---- synchronised BatchNorm statistics for data-parallel training -------------------------------------------------------
# Off by default ("local-BN DP": statistics per rank).  set_bn_sync(world, group) makes every training-mode BatchNorm of
# this module all-reduce its per-channel partial sums (sum, sum of squares; backward: sum dy, sum dy*xhat) over the
# ranks, i.e. the statistics of the GLOBAL batch as on one device (reference semantics): one small
# all-reduce (NSLOT x 2 x C doubles) per BatchNorm and direction.
_BN_SYNC = {"world": 1, "group": None}


def set_bn_sync(world=1, group=None):
    _BN_SYNC["world"], _BN_SYNC["group"] = int(world), group


def bn_sync_world():
    return _BN_SYNC["world"]


def bn_allreduce_(buf):
    import torch.distributed as dist
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=_BN_SYNC["group"])
    return buf


def bn_scale_shift(bn: nn.BatchNorm2d, stats, count, training):
    """BatchNorm folded to per-channel (scale, shift).  training: from the batch statistics in `stats`
    (running statistics updated in place, as nn.BatchNorm2d does); eval: from the running statistics.
    Returns (scale, shift, save_mean, save_invstd)."""
    L = _lib.lib()
    Cc = bn.num_features
    dev = bn.weight.device
    scale = torch.empty(Cc, dtype=torch.float32, device=dev)
    shift = torch.empty(Cc, dtype=torch.float32, device=dev)
    g = bn.weight.detach()
    bta = bn.bias.detach()
    if training:
        if Cc % 16:
            raise NotImplementedError("libsrbh BatchNorm statistics need a channel count that is a multiple of 16")
        mean = torch.empty(Cc, dtype=torch.float32, device=dev)
        invstd = torch.empty(Cc, dtype=torch.float32, device=dev)
        mom = 0.1 if bn.momentum is None else bn.momentum
        if _BN_SYNC["world"] > 1:
            bn_allreduce_(stats)
            count = count * _BN_SYNC["world"]
        rm = bn.running_mean.data_ptr() if bn.track_running_stats else None
        rv = bn.running_var.data_ptr() if bn.track_running_stats else None
        fin = L.srbh_bn_finalize_clear if stats_clean(stats) else L.srbh_bn_finalize
        _lib.check(fin(stats.data_ptr(), Cc, float(count), g.data_ptr(), bta.data_ptr(), bn.eps, mom,
                       rm, rv, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                       _lib.stream_ptr()), "bn_finalize")
        stats_release(stats)
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            note_batch(bn)
        return scale, shift, mean, invstd
    _lib.check(L.srbh_bn_eval_scale_shift(Cc, g.data_ptr(), bta.data_ptr(), bn.running_mean.data_ptr(),
                                          bn.running_var.data_ptr(), bn.eps, scale.data_ptr(), shift.data_ptr(),
                                          _lib.stream_ptr()), "bn_eval_scale_shift")
    return scale, shift, None, None


# ---- num_batches_tracked: ONE fused increment per forward instead of one launch per BatchNorm ---------------------------------
# nn.BatchNorm2d.forward does `num_batches_tracked.add_(1)` per module: 68 one-element kernels per training step of the height
# model, each costing a launch and ~40 us of host dispatch in a step that is bound by exactly that.  Training-mode BatchNorms
# (libsrbh's here, the stock-op ones in encoders.bn_act) note their counter; the outermost forward adds 1 to all of them with a
# single multi-tensor op.  The counters are exact whenever a forward has returned (state_dict / checkpoints see no difference).
_NBT = {"pending": [], "depth": 0}


def note_batch(bn):
    _NBT["pending"].append(bn.num_batches_tracked)


def flush_batches(force=False):
    if _NBT["pending"] and (force or _NBT["depth"] == 0):
        with torch.no_grad():
            torch._foreach_add_(_NBT["pending"], 1)
        _NBT["pending"] = []


class defer_batch_counters:
    """context of an outer forward: inner modules do not flush, the outermost exit does (once)"""

    def __enter__(self):
        _NBT["depth"] += 1

    def __exit__(self, *exc):
        _NBT["depth"] -= 1
        flush_batches()
        return False


RELU_BITS = _os.environ.get("SRBH_RELU_BITS", "1") == "1"     # training: the block-closing ReLU's pattern saved as bits for the backward


def bn_add_relu(a, sa, ha, idt, si=None, hi=None, want_bits=False):
    """out = relu(a * sa + ha + [idt * si + hi | idt]) (fp32 NHWC); `a` / `idt` may be fp16 tensors (the training step's saved activations).
    want_bits (training, C % 4 == 0): returns (out, bits) -- the ReLU's activity pattern, 1 bit per element, which the backward's reduce pass
    reads instead of the fp32 output (hrfuse_autograd.bn_backward(relu_ref=bits))."""
    B, Cc, H, W = a.shape
    out = empty_nhwc(B, Cc, H, W, a.device)
    io = (1 if a.dtype == torch.float16 else 0) | (2 if idt.dtype == torch.float16 else 0)
    if want_bits:
        L = _lib.lib()
        bits = torch.empty(L.srbh_relu_bits_bytes(B * H * W, Cc) // 8, dtype=torch.int64, device=a.device)
        _lib.check(L.srbh_bn_add_relu_bits(a.data_ptr(), sa.data_ptr(), ha.data_ptr(), idt.data_ptr(),
                                           None if si is None else si.data_ptr(), None if hi is None else hi.data_ptr(),
                                           out.data_ptr(), bits.data_ptr(), B * H * W, Cc, io, _lib.stream_ptr()), "bn_add_relu_bits")
        return out, bits
    _lib.check(_lib.lib().srbh_bn_add_relu_io(a.data_ptr(), sa.data_ptr(), ha.data_ptr(), idt.data_ptr(),
                                              None if si is None else si.data_ptr(), None if hi is None else hi.data_ptr(),
                                              out.data_ptr(), B * H * W, Cc, io, _lib.stream_ptr()), "bn_add_relu")
    return out


def _no_eager(name):
    raise RuntimeError(f"{name}: parameter container of the HIP head; it has no eager forward")


# ----------------------------------------------------------------------------- modules
def default_conv(in_channels, out_channels, kernel_size, bias=True):
    """reference SR/HRfuse.py:11-14"""
    return nn.Conv2d(in_channels, out_channels, kernel_size, padding=(kernel_size // 2), bias=bias)


def conv3x3(in_planes: int, out_planes: int, stride: int = 1, groups: int = 1, dilation: int = 1) -> nn.Conv2d:
    """reference SR/HRfuse.py:93-104"""
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, groups=groups, bias=False,
                     dilation=dilation)


def conv1x1(in_planes: int, out_planes: int, stride: int = 1) -> nn.Conv2d:
    """reference SR/HRfuse.py:106-108"""
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class Upsampler(nn.Sequential):
    """[conv3x3 n->4n (bias), PixelShuffle(2)] x log2(scale) (reference SR/HRfuse.py:17-44); Sequential keys
    0,2,.. hold the convs, odd indices the (parameter-free) PixelShuffle modules, exactly as upstream."""

    def __init__(self, conv=default_conv, scale=4, n_feats=16, bn=False, act=False, bias=True):
        m = []
        if (scale & (scale - 1)) == 0:
            for _ in range(int(math.log(scale, 2))):
                m.append(conv(n_feats, 4 * n_feats, 3, bias))
                m.append(nn.PixelShuffle(2))
                if bn:
                    m.append(nn.BatchNorm2d(n_feats))
                if act == "relu":
                    m.append(nn.ReLU(True))
                elif act == "prelu":
                    m.append(nn.PReLU(n_feats))
        elif scale == 3:
            m.append(conv(n_feats, 9 * n_feats, 3, bias))
            m.append(nn.PixelShuffle(3))
            if bn:
                m.append(nn.BatchNorm2d(n_feats))
            if act == "relu":
                m.append(nn.ReLU(True))
            elif act == "prelu":
                m.append(nn.PReLU(n_feats))
        else:
            raise NotImplementedError
        super().__init__(*m)
        self._hip_ok = (scale & (scale - 1)) == 0 and not bn and not act
        self._packs = {}

    def forward(self, x, out_h16=False):
        """out_h16 (inference chain, fp16-operand mode, n_feats == 16): the up-sampled tensor -- and the intermediate one -- as fp16 NHWC:
        the PixelShuffle store rounds once where the consuming conv's staging would (same numbers), and every tensor written here is read
        again at half the bytes"""
        _require_dev(x, "Upsampler")
        if not self._hip_ok:
            raise NotImplementedError("libsrbh Upsampler supports power-of-two scales without bn/act "
                                      "(the only configuration the reference instantiates)")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from . import hrfuse_autograd as AG
            return AG.upsampler_forward(self, x)
        x = to_nhwc(x)
        h16 = bool(out_h16) and fp16_chain(self) and all(m.out_channels == 64 for m in self if isinstance(m, nn.Conv2d))
        for i, mod in enumerate(self):
            if isinstance(mod, nn.Conv2d):
                x, _ = hconv([x], mod, self._packs.setdefault(i, _PackedConv()), ps2=True, out_h16=h16)
        return x


class BasicBlock(nn.Module):
    """ResNet basic block with the reference's constructor (SR/HRfuse.py:109-159)."""

    def __init__(self, inplanes: int, planes: int, stride: int = 1, groups: int = 1, base_width: int = 64,
                 dilation: int = 1, norm_layer: Optional[Callable[..., nn.Module]] = None, expansion: int = 1) -> None:
        super().__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = None
        self.stride = stride
        if stride != 1 or inplanes != planes * expansion:
            self.downsample = nn.Sequential(conv1x1(inplanes, planes * expansion, stride), norm_layer(planes * expansion))
        self._p1, self._p2, self._pd = _PackedConv(), _PackedConv(), _PackedConv()

    def _check(self):
        if self.stride != 1:
            raise NotImplementedError("libsrbh BasicBlock supports stride 1 (all the reference's call sites)")
        for bn in (self.bn1, self.bn2):
            if not isinstance(bn, nn.BatchNorm2d):
                raise NotImplementedError("libsrbh BasicBlock supports nn.BatchNorm2d norm layers")

    def forward_nhwc(self, srcs, out_h16=False):
        """srcs: list of 1..2 NHWC tensors whose channel concat is the block input (no autograd).
        Inference with fp16 operands (`head_h16()`): the block's intermediates live in memory as fp16 -- bn1 + ReLU run in conv1's
        epilogue (the value conv2 would round while staging is rounded there, once: same numbers), the downsample branch and,
        with out_h16, the block output are written as fp16 too (the residual stream of a chain of blocks is then fp16: one extra
        rounding of 2^-11 per block on the identity path).  Every conv then moves 32 + 32 instead of 64 + 64 bytes per pixel."""
        self._check()
        tr = self.training
        B, _, H, W = srcs[0].shape
        n = B * H * W
        if (not tr and FP16_ACTIVATIONS and head_h16() and self.bn2.num_features % 16 == 0 and self.bn1.num_features % 4 == 0
                and all(t.shape[1] % 4 == 0 for t in srcs)):
            s1, h1, _, _ = bn_scale_shift(self.bn1, None, n, False)
            s2, h2, _, _ = bn_scale_shift(self.bn2, None, n, False)
            if self.downsample is not None and len({t.dtype for t in srcs}) == 1:
                # conv1 + the 1x1 downsample conv: one pass over the input (sources of ONE element type: all fp32, or -- round 4: the fp16
                # feature hand-off -- all fp16; a mixed pair takes the two launches below)
                sd, hd, _, _ = bn_scale_shift(self.downsample[1], None, n, False)
                a1, _, idt, _ = hconv_entry(srcs, self.conv1, self._p1, self.downsample[0], self._pd, postd=(sd, hd), post1=(s1, h1),
                                            post1_relu=True, out_h16=True)
            elif self.downsample is not None:
                a1, _ = hconv(srcs, self.conv1, self._p1, post=(s1, h1), post_relu=True, out_h16=True)
                sd, hd, _, _ = bn_scale_shift(self.downsample[1], None, n, False)
                idt, _ = hconv(srcs, self.downsample[0], self._pd, post=(sd, hd), out_h16=True)
            else:
                if len(srcs) != 1:
                    raise ValueError("identity path needs a single source")
                x0 = srcs[0]
                if (HBLOCK16 and x0.dtype == torch.float16 and x0.shape[1] == 16 and self.bn1.num_features == 16 and self.bn2.num_features == 16
                        and _lib.lib().srbh_hblock16_supported(H, W)):
                    # the whole plain block as ONE pass (srbh_hblock16_eval): a1 never leaves the compute unit; same bits as the two launches below
                    return hblock16(x0, self, (s1, h1), (s2, h2), out_h16)
                a1, _ = hconv(srcs, self.conv1, self._p1, post=(s1, h1), post_relu=True, out_h16=True)
                idt = x0
            out, _ = hconv([a1], self.conv2, self._p2, post=(s2, h2), res=idt, post_relu=True, out_h16=out_h16)
            return out
        if any(t.dtype != torch.float32 for t in srcs):
            raise RuntimeError("libsrbh BasicBlock: fp16 activations outside the fp16 inference chain")
        fuse_entry = self.downsample is not None and head_h16()      # conv1 + the 1x1 downsample conv: one pass over the input
        infer = not tr and self.bn2.num_features % 16 == 0
        d = std = None
        if fuse_entry:
            sd = hd = None
            if infer:
                sd, hd, _, _ = bn_scale_shift(self.downsample[1], None, n, False)
            c1, st1, d, std = hconv_entry(srcs, self.conv1, self._p1, self.downsample[0], self._pd, want_stats=tr,
                                          postd=(sd, hd) if infer else None)
        else:
            c1, st1 = hconv(srcs, self.conv1, self._p1, want_stats=tr)
        s1, h1, _, _ = bn_scale_shift(self.bn1, st1, n, tr)
        if infer:
            # inference: BatchNorm is a per-channel affine -> bn2, the skip connection and the final ReLU run in conv2's
            # epilogue (and the downsample BatchNorm in the 1x1 conv's): no c2 round trip, no separate elementwise pass
            s2, h2, _, _ = bn_scale_shift(self.bn2, None, n, False)
            if fuse_entry:
                idt = d
            elif self.downsample is not None:
                sd, hd, _, _ = bn_scale_shift(self.downsample[1], None, n, False)
                idt, _ = hconv(srcs, self.downsample[0], self._pd, post=(sd, hd))
            else:
                if len(srcs) != 1:
                    raise ValueError("identity path needs a single source")
                idt = srcs[0]
            out, _ = hconv([c1], self.conv2, self._p2, pre=(s1, h1, True), post=(s2, h2), res=idt, post_relu=True)
            return out
        c2, st2 = hconv([c1], self.conv2, self._p2, pre=(s1, h1, True), want_stats=tr)
        s2, h2, _, _ = bn_scale_shift(self.bn2, st2, n, tr)
        if self.downsample is not None:
            if not fuse_entry:
                d, std = hconv(srcs, self.downsample[0], self._pd, want_stats=tr)
            sd, hd, _, _ = bn_scale_shift(self.downsample[1], std, n, tr)
            return bn_add_relu(c2, s2, h2, d, sd, hd)
        if len(srcs) != 1:
            raise ValueError("identity path needs a single source")
        return bn_add_relu(c2, s2, h2, srcs[0])

    def forward(self, x):
        _require_dev(x, "BasicBlock")
        return run_blocks([self], [x])


def _needs_grad(mods, tensors):
    if not torch.is_grad_enabled():
        return False
    return any(t.requires_grad for t in tensors) or any(p.requires_grad for m in mods for p in m.parameters())


def run_blocks(blocks, inputs, out_h16=False):
    """Run a chain of BasicBlocks on the channel concat of `inputs` ((B,C,H,W) tensors).  out_h16: the caller consumes the result
    through another libsrbh conv and takes it as an fp16 NHWC tensor (inference chain, see BasicBlock.forward_nhwc)."""
    if _needs_grad(blocks, inputs):
        from . import hrfuse_autograd as AG
        out = AG.blocks_forward(blocks, inputs)
        flush_batches()
        return out
    x = [to_nhwc(t) for t in inputs]
    for i, b in enumerate(blocks):
        x = [b.forward_nhwc(x, out_h16 or i + 1 < len(blocks))]      # (a non-chain block ignores the flag and returns fp32)
    flush_batches()
    return x[0]


class HRfeature(nn.Sequential):
    """Three BasicBlocks in_chans -> mid -> mid -> out (reference SR/HRfuse.py:164-169)."""

    def __init__(self, in_chans, mid_chans=64, out_chans=64):
        super().__init__(BasicBlock(in_chans, mid_chans, stride=1), BasicBlock(mid_chans, mid_chans, stride=1),
                         BasicBlock(mid_chans, out_chans, stride=1))

    def forward(self, x, out_h16=False):
        # (x may be RRDBNet.forward_feature(..., out_dtype=float16): an fp16 channels_last tensor, staged verbatim by the fp16-operand
        #  kernels -- inference chain and the 'f16' training mode; the exact-fp32 mode refuses it)
        _require_dev(x, "HRfeature", h16_ok=head_h16())
        if x.dtype == torch.float16 and to_nhwc(x) is not x:
            raise ValueError("HRfeature: an fp16 input must be channels_last (RRDBNet.forward_feature(out_dtype=torch.float16))")
        return run_blocks(list(self), [x], out_h16 and fp16_chain(self))


class _LastConv:
    """conv_last helper shared by the fuse heads."""

    @staticmethod
    def run(mod, conv, x):
        if _needs_grad([conv], [x]):
            from . import hrfuse_autograd as AG
            return AG.conv_forward(conv, mod._plast, x)
        out, _ = hconv([to_nhwc(x)], conv, mod._plast)
        return out


class HRfuse_residual(nn.Module):
    """upsample x_lr x4, concat with x_hr, three BasicBlocks, conv_last (reference SR/HRfuse.py:173-190)."""

    def __init__(self, hr_chans=16, lr_chans=16, mid_chans=16, out_chans=3, upscale=4):
        super().__init__()
        self.upsampler = Upsampler(scale=upscale, n_feats=lr_chans)
        self.fuse = nn.Sequential(BasicBlock(hr_chans + lr_chans, mid_chans, stride=1),
                                  BasicBlock(mid_chans, mid_chans, stride=1),
                                  BasicBlock(mid_chans, mid_chans, stride=1))
        self.conv_last = nn.Conv2d(mid_chans, out_chans, 3, 1, 1)
        self._plast = _PackedConv()

    def forward(self, x_lr, x_hr):
        _require_dev(x_lr, "HRfuse_residual")
        _require_dev(x_hr, "HRfuse_residual", h16_ok=fp16_chain(self))      # (HRfeature(..., out_h16=True) inside the inference chain)
        x_lr = self.upsampler(x_lr, out_h16=x_hr.dtype == torch.float16)    # (same element type as x_hr: the fused fp16 entry kernel)
        x = run_blocks(list(self.fuse), [x_lr, x_hr], fp16_chain(self))      # (conv_last reads the chain's fp16 output)
        return _LastConv.run(self, self.conv_last, x)


class Refine_residual(nn.Module):
    """concat, three BasicBlocks, conv_last -- no upsampler (reference SR/HRfuse.py:217-228)."""

    def __init__(self, hr_chans=16, lr_chans=16, mid_chans=16, out_chans=3):
        super().__init__()
        self.fuse = nn.Sequential(BasicBlock(hr_chans + lr_chans, mid_chans, stride=1),
                                  BasicBlock(mid_chans, mid_chans, stride=1),
                                  BasicBlock(mid_chans, mid_chans, stride=1))
        self.conv_last = nn.Conv2d(mid_chans, out_chans, 3, 1, 1)
        self._plast = _PackedConv()

    def forward(self, x_lr, x_hr):
        _require_dev(x_lr, "Refine_residual")
        x = run_blocks(list(self.fuse), [x_lr, x_hr])
        return _LastConv.run(self, self.conv_last, x)


class HRupsample(nn.Module):
    """upsampler + conv_last (reference SR/HRfuse.py:193-202)."""

    def __init__(self, lr_chans=16, out_chans=3, upscale=4):
        super().__init__()
        self.upsampler = Upsampler(scale=upscale, n_feats=lr_chans)
        self.conv_last = nn.Conv2d(lr_chans, out_chans, 3, 1, 1)
        self._plast = _PackedConv()

    def forward(self, x):
        _require_dev(x, "HRupsample")
        return _LastConv.run(self, self.conv_last, self.upsampler(x))


class GeoNet(nn.Module):
    """three BasicBlocks under ``feat`` (reference SR/HRfuse.py:205-214)."""

    def __init__(self, in_chans=4, mid_chans=16):
        super().__init__()
        self.feat = nn.Sequential(BasicBlock(in_chans, mid_chans, stride=1), BasicBlock(mid_chans, mid_chans, stride=1),
                                  BasicBlock(mid_chans, mid_chans, stride=1))

    def forward(self, x):
        _require_dev(x, "GeoNet")
        return run_blocks(list(self.feat), [x])


class _ConvBnReluFuse(nn.Module):
    """shared body of HRfuse / HRfuse_x2: conv-BN-ReLU x2 as an nn.Sequential named ``fuse`` with the reference's
    integer keys (0,1,3,4 hold parameters), an Upsampler and conv_last (reference SR/HRfuse.py:47-90)."""

    def __init__(self, hr_channel, lr_channel, mid_channel, out_channel, upscale):
        super().__init__()
        self.fuse = nn.Sequential(
            nn.Conv2d(hr_channel + lr_channel, mid_channel, 3, 1, 1, bias=False), nn.BatchNorm2d(mid_channel),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channel, mid_channel, 3, 1, 1, bias=False), nn.BatchNorm2d(mid_channel),
            nn.ReLU(inplace=True))
        self.upsampler = Upsampler(scale=upscale, n_feats=mid_channel)
        self.conv_last = nn.Conv2d(mid_channel, out_channel, 3, 1, 1)
        self._pf0, self._pf3, self._plast = _PackedConv(), _PackedConv(), _PackedConv()

    def _fuse_nhwc(self, srcs):
        if _needs_grad([self], srcs):
            raise NotImplementedError("HRfuse / HRfuse_x2 are inference-only in libsrbh (they are not instantiated by "
                                      "the reference's train.py / predict scripts)")
        tr = self.training
        B, _, H, W = srcs[0].shape
        c1, st1 = hconv(srcs, self.fuse[0], self._pf0, want_stats=tr)
        s1, h1, _, _ = bn_scale_shift(self.fuse[1], st1, B * H * W, tr)
        c2, st2 = hconv([c1], self.fuse[3], self._pf3, pre=(s1, h1, True), want_stats=tr)
        s2, h2, _, _ = bn_scale_shift(self.fuse[4], st2, B * H * W, tr)
        return c2, (s2, h2, True)   # the trailing BN+ReLU is folded into whichever conv consumes c2


class HRfuse(_ConvBnReluFuse):
    """fuse at low resolution, then upsample (reference SR/HRfuse.py:47-65)."""

    def __init__(self, hr_channel=16, lr_channel=16, mid_channel=16, out_channel=3, upscale=4):
        super().__init__(hr_channel, lr_channel, mid_channel, out_channel, upscale)

    def forward(self, x_lr, x_hr):
        _require_dev(x_lr, "HRfuse")
        c2, pre = self._fuse_nhwc([to_nhwc(x_lr), to_nhwc(x_hr)])
        x = c2
        first = True
        for i, mod in enumerate(self.upsampler):
            if isinstance(mod, nn.Conv2d):
                x, _ = hconv([x], mod, self.upsampler._packs.setdefault(i, _PackedConv()), pre=pre if first else None,
                             ps2=True)
                first = False
        out, _ = hconv([x], self.conv_last, self._plast)
        return out


class HRfuse_x2(_ConvBnReluFuse):
    """upsample x_lr first, fuse at high resolution (reference SR/HRfuse.py:68-90)."""

    def __init__(self, hr_channel=16, lr_channel=16, mid_channel=16, out_channel=3, upscale=4):
        super().__init__(hr_channel, lr_channel, mid_channel, out_channel, upscale)

    def forward(self, x_lr, x_hr):
        _require_dev(x_lr, "HRfuse_x2")
        with torch.no_grad():
            up = self.upsampler(x_lr)
        c2, pre = self._fuse_nhwc([up, to_nhwc(x_hr)])
        out, _ = hconv([c2], self.conv_last, self._plast, pre=pre)
        return out
