"""Per-call device timing + ALGORITHMIC bytes of the libsrbh calls of one step: the whole-head HBM roofline (`bench.py`
`train_step.head_roofline`; round-2 VERDICT: "a whole-head roofline, not three cherry-picked kernels").

`with KernelProfile() as kp: step()` wraps every exported libsrbh function that launches head / loss work: each call is bracketed
by two HIP events on the current stream (so a call that issues several kernels -- the weight gradient: partial sums + two ordered
reduce launches -- is timed as one unit) and priced with the bytes its arguments imply: every tensor it reads or writes, once, at the
element size its flags say (the 16-bit forms are priced at 2 bytes per element).  `kp.table()` groups the calls by (entry point,
shape) and reports calls, microseconds, bytes and the fraction of the HBM peak.  Events serialise nothing, but they do put a marker
between launches; the step as a whole runs ~2 % slower under the profile, so the headline is never measured under it.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import torch

from . import _lib

PEAK_HBM_GBS = 8000.0


def _px(a):
    return a.B * a.H * a.W


def _hconv_bytes(a, opt16):
    """srbh_hconv_args: sources + output (+ residual) once; 16-bit tensors (io_h16) at 2 bytes"""
    io = a.io_h16
    e = lambda bit: 2 if (io & bit) else 4       # noqa: E731
    px = _px(a)
    n = px * (a.c0 * e(1) + a.c1 * e(2))
    n += px * a.cout * e(8)                       # (PixelShuffle store: same element count)
    if a.res1:
        n += px * a.cout * e(4)
    if a.res2:
        n += px * a.cout * 4
    if getattr(a, "bstat_c", None):               # backward-statistics epilogue: one more fp32 tensor read
        n += px * a.cout * 4
    return n


def _desc_hconv(name, a):
    t = f"{name} {a.c0}{'+' + str(a.c1) if a.c1 else ''}->{a.cout} k{a.ksize}"
    if a.pixelshuffle2:
        t += " ps2"
    if a.stats:
        t += " +bwd-stats" if getattr(a, "bstat_c", None) else " +stats"
    if a.res1:
        t += " +res"
    if a.io_h16:
        t += f" io16={a.io_h16}"
    return t + f" @{a.H}x{a.W}"


def _model(name, args):
    """-> (description, algorithmic bytes) of one call, or None for entry points that are not priced"""
    if name == "srbh_hconv_f32":
        a = args[0]._obj
        return _desc_hconv("hconv_f32", a), _hconv_bytes(a, False)
    if name == "srbh_hconv_h16":
        a = args[0]._obj
        return _desc_hconv("hconv_bf16" if args[1] else "hconv_fp16", a), _hconv_bytes(a, True)
    if name == "srbh_hconv_entry_h16":
        a, d = args[0]._obj, args[1]._obj
        # one pass over the input when fused (the common case); priced as such: sources once, two outputs
        e = 2 if (a.io_h16 & 8) else 4
        px = _px(a)
        return (f"hconv_entry {a.c0}{'+' + str(a.c1) if a.c1 else ''}->16 k3 + 16 k1{' +stats' if a.stats else ''}{' src16' if a.io_h16 & 3 else ''}{' out16' if a.io_h16 & 8 else ''} @{a.H}x{a.W}",
                px * (a.c0 * (2 if a.io_h16 & 1 else 4) + a.c1 * (2 if a.io_h16 & 2 else 4) + 2 * 16 * e))
    if name == "srbh_hconv_wgrad_entry_b16":
        a, d = args[0]._obj, args[1]._obj
        px = _px(a)
        ex = 2 if (a.io & 1) else 4
        ed = 2 if (a.io & 2) else 4
        # one pass over the input when fused (the common case): sources once, two dY tensors
        return (f"wgrad_entry_bf16 {a.c0}{'+' + str(a.c1) if a.c1 else ''}->{a.cout} k3 + k1{' io=' + str(a.io) if a.io else ''} @{a.H}x{a.W}",
                px * (a.c0 * ex + a.c1 * 4 + 2 * a.cout * ed))
    if name == "srbh_hbwd16":
        a = args[0]._obj
        px = a.B * a.H * a.W
        # each tensor once at its stored element size: g bf16, c / x fp32, dx bf16 | fp32, the statistics epilogue's c fp32 or the skip gradient bf16
        nbytes = px * 16 * (2 + 4 + 4 + (2 if a.dx_b16 else 4) + (4 if a.stats else 0) + (2 if a.res else 0))
        return (f"hbwd16 (BN apply + wgrad + dgrad, one pass) 16->16 k3{' +bwd-stats' if a.stats else ''}{' +mask' if a.mask_scale else ''}{' +res' if a.res else ''} "
                f"out={'bf16' if a.dx_b16 else 'fp32'} @{a.H}x{a.W}", nbytes)
    if name in ("srbh_hconv_wgrad_f32", "srbh_hconv_wgrad_b16"):
        a = args[0]._obj
        px = _px(a)
        ex = 2 if (a.io & 1) else 4
        ed = 2 if (a.io & 2) else 4
        return (f"{'wgrad_bf16' if name.endswith('b16') else 'wgrad_f32'} {a.c0}{'+' + str(a.c1) if a.c1 else ''}->{a.cout} k{a.ksize}{' io=' + str(a.io) if a.io else ''} @{a.H}x{a.W}",
                px * (a.c0 * ex + a.c1 * 4 + a.cout * ed))
    if name == "srbh_bn_add_relu":
        npix, Cc = args[7], args[8]
        return f"bn_add_relu C={Cc}", npix * Cc * 12
    if name == "srbh_bn_add_relu_io":
        npix, Cc, io = args[7], args[8], args[9]
        return f"bn_add_relu C={Cc}{' io=' + str(io) if io else ''}", npix * Cc * ((2 if io & 1 else 4) + (2 if io & 2 else 4) + 4)
    if name == "srbh_bn_add_relu_bits":      # (+ the ReLU's activity pattern, 1 bit per element)
        npix, Cc, io = args[8], args[9], args[10]
        return f"bn_add_relu+bits C={Cc}{' io=' + str(io) if io else ''}", npix * Cc * ((2 if io & 1 else 4) + (2 if io & 2 else 4) + 4) + npix * Cc // 8
    if name == "srbh_bn_bwd_reduce":
        g, c, npix, Cc = args[0], args[1], args[6], args[7]
        return f"bn_bwd_reduce C={Cc}{'' if c else ' (bias grad)'}", npix * Cc * (4 + (4 if c else 0))
    if name == "srbh_bn_bwd_reduce_relu":
        npix, Cc = args[6], args[7]
        return f"bn_bwd_reduce+relu C={Cc}", npix * Cc * (4 + 4 + (4 if args[2] else 0) + (4 if args[3] else 0))
    if name == "srbh_bn_bwd_reduce_io":
        relu_ref, dz, c, npix, Cc, io = args[1], args[2], args[3], args[8], args[9], args[11]
        n = (2 if io & 4 else 4) + (4 if relu_ref else 0) + ((2 if io & 1 else 4) if dz else 0) + ((2 if io & 2 else 4) if c else 0)
        return f"bn_bwd_reduce{'+relu' if relu_ref else ''} C={Cc} io={io}", npix * Cc * n
    if name == "srbh_bn_bwd_apply":
        npix, Cc = args[10], args[11]
        return f"bn_bwd_apply C={Cc}", npix * Cc * 12
    if name == "srbh_bn_bwd_apply_io":
        npix, Cc, io = args[10], args[11], args[12]
        return f"bn_bwd_apply C={Cc} io={io}", npix * Cc * ((2 if io & 4 else 4) + (2 if io & 2 else 4) + (2 if io & 1 else 4))
    if name == "srbh_relu_mask_mul":
        return "relu_mask_mul", args[3] * 12
    if name == "srbh_add_inplace":
        return "add_inplace", args[2] * 12
    if name == "srbh_ps2_inverse":
        B, H, W, Cc = args[2], args[3], args[4], args[5]
        return f"ps2_inverse C={Cc} @{H}x{W}", B * H * W * 4 * Cc * 8
    if name == "srbh_nchw_to_nhwc_f32":
        B, Cc, H, W = args[2], args[3], args[4], args[5]
        return f"nchw_to_nhwc C={Cc} @{H}x{W}", B * Cc * H * W * 8
    if name in ("srbh_bn_finalize", "srbh_bn_bwd_finalize", "srbh_bn_finalize_clear", "srbh_bn_bwd_finalize_clear", "srbh_bn_eval_scale_shift"):
        return name[5:], 0
    if name == "srbh_wmse_sum":
        return "loss wmse_sum", args[3] * 12
    if name == "srbh_wmse_grad":
        return "loss wmse_grad", args[3] * 16
    if name == "srbh_cedice_sums":
        Bn, Cc, hw = args[1], args[2], args[3]
        return f"loss cedice_sums C={Cc}", Bn * hw * (4 * Cc + 12)           # logits + int64 labels + weights
    if name == "srbh_cedice_grad":
        Bn, Cc, hw = args[1], args[2], args[3]
        return f"loss cedice_grad C={Cc}", Bn * hw * (8 * Cc + 12)
    return _model_encdec(name, args)


def _model_encdec(name, args):
    """encoder / decoder calls (group "encdec"): planes of 2x2 .. 64x64 -- latency chains, priced with their bytes all the same"""
    if name in ("srbh_bn_act_train_fwd", "srbh_bn_act_train_bwd"):
        a = args[0]._obj
        n = a.B * a.C * a.HW * 4
        if name.endswith("fwd"):
            return (f"bn_act_train_fwd act={a.act}{' +res' if a.res else ''}{' +pool' if a.pooled else ''} C={a.C} HW={a.HW}", n * (3 if a.res else 2))
        return f"bn_act_train_bwd act={a.act}{' +gate' if a.gate else ''} C={a.C} HW={a.HW}", n * (3 if a.dx else 2)
    if name == "srbh_se_train_fwd":
        B, Cc, SQ, HW = args[9], args[10], args[11], args[12]
        return f"se_train_fwd C={Cc} SQ={SQ} HW={HW}", B * Cc * HW * 8 + 2 * Cc * SQ * 4
    if name == "srbh_se_train_bwd":
        B, Cc, SQ, HW = args[18], args[19], args[20], args[21]
        return f"se_train_bwd C={Cc} SQ={SQ} HW={HW}", B * Cc * HW * 8 + 4 * Cc * SQ * 4
    if name in ("srbh_pwconv_fwd", "srbh_pwconv_fwd_wt", "srbh_pwconv_bwd_data"):
        B, Cin, Cout, HW = args[3], args[4], args[5], args[6]
        return f"{name[5:]} {Cin}->{Cout} HW={HW}", (B * HW * (Cin + Cout) + Cin * Cout) * 4
    if name == "srbh_pwconv_bwd_weight":
        B, Cin, Cout, HW = args[4], args[5], args[6], args[7]
        return f"pwconv_bwd_weight {Cin}->{Cout} HW={HW}", (B * HW * (Cin + Cout) + Cin * Cout) * 4
    if name in ("srbh_dwconv_fwd", "srbh_dwconv_bwd_data"):
        B, Cc, Hh, Ww, K, st, OH, OW = args[3], args[4], args[5], args[6], args[7], args[8], args[11], args[12]
        return f"{name[5:]} C={Cc} k{K} s{st} @{Hh}x{Ww}", B * Cc * (Hh * Ww + OH * OW) * 4
    if name == "srbh_dwconv_bwd_weight":
        B, Cc, Hh, Ww, K, st, OH, OW = args[4], args[5], args[6], args[7], args[8], args[9], args[12], args[13]
        return f"dwconv_bwd_weight C={Cc} k{K} s{st} @{Hh}x{Ww}", B * Cc * (Hh * Ww + OH * OW) * 4
    if name in ("srbh_up2_cat_fwd", "srbh_up2_cat_bwd"):
        B, Cx, Cs, Hh, Ww = args[3], args[4], args[5], args[6], args[7]
        return f"{name[5:]} {Cx}+{Cs} @{Hh}x{Ww}", B * (Cx * Hh * Ww + (Cx + 2 * Cs) * 4 * Hh * Ww) * 4
    if name == "srbh_transpose_many":
        return "transpose_many (all 1x1 weights)", 0
    if name in ("srbh_mbconv_mid_fwd", "srbh_mbconv_mid_bwd"):          # BatchNorm0 + SiLU -> depthwise -> BatchNorm1 + SiLU (+ pool), one launch
        a = args[0]._obj
        n = a.B * a.C * a.H * a.W * 4
        # forward: e_pre read, d_pre + y written; backward: dout, d_pre, e_pre read, de_pre written
        return f"{name[5:]} C={a.C} k{a.K} @{a.H}x{a.W}", n * (3 if name.endswith("fwd") else 4)
    if name == "srbh_dconv_fwd":
        B, Cin, Cout, Hh, Ww, b16 = args[3], args[4], args[5], args[6], args[7], args[8]
        return f"dconv_{'dgrad' if b16 else 'fwd'} {Cin}->{Cout} @{Hh}x{Ww}", (B * Hh * Ww * (Cin + Cout) + 9 * Cin * Cout) * 4
    if name == "srbh_dconv_wgrad":
        B, Cin, Cout, Hh, Ww = args[4], args[5], args[6], args[7], args[8]
        return f"dconv_wgrad {Cin}->{Cout} @{Hh}x{Ww}", (B * Hh * Ww * (Cin + Cout) + 9 * Cin * Cout) * 4
    return None


class KernelProfile:
    def __init__(self, group="head"):
        """group: "head" = the head / loss entry points (the whole-head roofline); "encdec" = the encoder / decoder entry points"""
        self.calls = []          # (desc, bytes, ev0, ev1)
        self._saved = {}
        self.group = group

    def __enter__(self):
        L = _lib.lib()
        for name in _lib.SIGNATURES:
            fn = getattr(L, name)
            if _model_names(name) if self.group == "head" else _encdec_names(name):
                self._saved[name] = fn
                setattr(L, name, self._wrap(name, fn))
        return self

    def _wrap(self, name, fn):
        def call(*args):
            m = _model(name, args)
            if m is None:
                return fn(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            self.calls.append((m[0], m[1], e0, e1))
            return rc
        return call

    def __exit__(self, *exc):
        L = _lib.lib()
        for name, fn in self._saved.items():
            setattr(L, name, fn)
        self._saved = {}
        return False

    def table(self, steps=1):
        """-> (rows sorted by time, totals).  Row: kernel (entry point + shape), calls per step, us per call, algorithmic MB per
        call, GB/s, fraction of the HBM peak, ms per step."""
        torch.cuda.synchronize()
        agg = OrderedDict()
        for desc, nbytes, e0, e1 in self.calls:
            a = agg.setdefault(desc, [0, 0.0, 0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            a[2] += nbytes
        rows = []
        for desc, (n, ms, nbytes) in agg.items():
            rows.append({"kernel": desc, "calls_per_step": round(n / steps, 2), "us_per_call": round(ms / n * 1e3, 1),
                         "algorithmic_MB_per_call": round(nbytes / n / 1e6, 1),
                         "achieved_GBs": round(nbytes / ms / 1e6, 1) if ms > 0 and nbytes else None,
                         "frac_hbm_peak": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 4) if ms > 0 and nbytes else None,
                         "ms_per_step": round(ms / steps, 3)})
        rows.sort(key=lambda r: -r["ms_per_step"])
        tot_ms = sum(r["ms_per_step"] for r in rows)
        tot_b = sum(nb for _, (n, ms, nb) in agg.items()) / steps
        totals = {"ms_per_step": round(tot_ms, 3), "algorithmic_GB_per_step": round(tot_b / 1e9, 3),
                  "achieved_GBs": round(tot_b / tot_ms / 1e6, 1) if tot_ms else None,
                  "frac_hbm_peak": round(tot_b / tot_ms / 1e6 / PEAK_HBM_GBS, 4) if tot_ms else None,
                  "ideal_ms_at_peak": round(tot_b / PEAK_HBM_GBS / 1e6, 3)}
        return rows, totals


def _model_names(name):
    if name.startswith("srbh_bn_act_train"):       # (encoder / decoder BatchNorm: group "encdec")
        return False
    return name.startswith(("srbh_hconv_f32", "srbh_hconv_h16", "srbh_hconv_entry", "srbh_hconv_wgrad", "srbh_hbwd16", "srbh_bn_", "srbh_relu_mask",
                            "srbh_add_inplace", "srbh_ps2_inverse", "srbh_nchw_to_nhwc", "srbh_wmse", "srbh_cedice"))


def _encdec_names(name):
    return (name.startswith(("srbh_bn_act_train_fwd", "srbh_bn_act_train_bwd", "srbh_se_train_fwd", "srbh_se_train_bwd", "srbh_pwconv_fwd",
                             "srbh_pwconv_bwd_data", "srbh_pwconv_bwd_weight", "srbh_dwconv_fwd", "srbh_dwconv_bwd_data", "srbh_dwconv_bwd_weight",
                             "srbh_up2_cat", "srbh_transpose_many", "srbh_mbconv_mid_fwd", "srbh_mbconv_mid_bwd", "srbh_dconv_fwd", "srbh_dconv_wgrad"))
            and name != "srbh_dconv_fwd_epi" and not name.endswith(("_supported", "_ws_bytes", "_ws_floats", "_splits")))
