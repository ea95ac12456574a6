"""ctypes binding of libsrbh.so (the C ABI declared in include/srbh.h) and its in-tree build recipe.

The library is plain HIP behind ``extern "C"`` (no torch types); PyTorch only supplies device
memory and the stream.  There is deliberately NO fallback: if the shared object is missing or a
call fails, a RuntimeError is raised (SURVEY.md 8b "Errors").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_PATH = os.path.join(_HERE, "libsrbh.so")
_DEV_LIB = os.environ.get("SRBH_LIB_PATH")      # developer A/B only (tools/ab_variants.sh): load another build of the same ABI
SOURCES = ["srbh_conv3x3.hip", "srbh_aux.hip", "srbh_rrdbnet.hip", "srbh_ptrunk.hip", "srbh_trunk_wgrad.hip", "srbh_ptail.hip", "srbh_head.hip", "srbh_head_bwd.hip", "srbh_mosaic.hip", "srbh_loader.hip", "srbh_loss.hip", "srbh_dwconv.hip", "srbh_mbconv.hip", "srbh_pwconv.hip", "srbh_dconv.hip", "srbh_optim.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# NOTE: `-mllvm -amdgpu-mfma-vgpr-form=1` (accumulators in VGPRs: no v_accvgpr copies at K-loop back-edges).  Round 1 saw the
# first persistent trunk kernel produce non-deterministic garbage with it; round 3 re-ran it on the current kernel (ptrunk3:
# fully unrolled steps, every LDS-DMA guarded by an explicit vmcnt wait -- the round-1 kernel lacked those waits, DESIGN.md 5.1
# "lessons"): tests/test_gpu_rrdbnet.py 13/13 green, parity 7.095e-4 identical, hazcheck clean -- and NO speed-up (3.891 vs 3.897 ms,
# 3.911 vs 3.940 ms per trunk launch on one box; the compiler then parks the long-lived residual stream in AGPRs instead: 5 914 vs
# 5 486 v_accvgpr moves in the ISA).  The garbage was the missing waits, not the flag; the flag buys nothing here and stays off.
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-inline-asm"]   # (M0 is clobbered on purpose by the LDS-DMA asm)
if os.environ.get("SRBH_HIPFLAGS_OVERRIDE"):      # developer bisecting only
    HIPFLAGS = os.environ["SRBH_HIPFLAGS_OVERRIDE"].split()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into the in-tree libsrbh.so (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "srbh.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    objs, procs = [], []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        cmd = [HIPCC, *HIPFLAGS, "-I", INCLUDE, "-I", CSRC, "-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out:
            print(out)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return LIB_PATH


PATH_NAMES = ("hconv16", "hconv_template", "entry_fused", "entry_split", "wgrad16", "wgrad_b16_generic", "wgrad_f32", "wgrad_entry_fused",
              "wgrad_entry_split", "hconv_up", "hbwd16", "hblock16")


def path_counters(reset=False):
    """{form: launches since the last reset} of the head entry points that choose between kernel forms (include/srbh.h SRBH_PATH_*)"""
    buf = (C.c_ulonglong * len(PATH_NAMES))()
    lib().srbh_path_counters(buf, len(PATH_NAMES), int(reset))
    return {k: int(v) for k, v in zip(PATH_NAMES, buf)}


# ---- C structs (mirror include/srbh.h) ------------------------------------------------------------
class ConvArgs(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("in_chunks_total", C.c_int), ("in_chunk0", C.c_int), ("in_chunks", C.c_int),
        ("w", C.c_void_p), ("bias", C.c_void_p), ("cout", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("upsample2x", C.c_int), ("lrelu", C.c_int),
        ("res_scale", C.c_float), ("res1", C.c_void_p), ("res1_update", C.c_int),
        ("res2_scale", C.c_float), ("res2", C.c_void_p), ("res2_update", C.c_int),
        ("skip", C.c_void_p),
        ("out16", C.c_void_p), ("out16_chunks_total", C.c_int), ("out16_chunk0", C.c_int),
        ("out32", C.c_void_p), ("out32_c", C.c_int), ("out16_nhwc", C.c_int),
    ]


class HConvArgs(C.Structure):
    _fields_ = [
        ("src0", C.c_void_p), ("c0", C.c_int),
        ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p), ("pre_relu", C.c_int),
        ("src1", C.c_void_p), ("c1", C.c_int),
        ("w", C.c_void_p), ("bias", C.c_void_p), ("cout", C.c_int), ("ksize", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("pixelshuffle2", C.c_int),
        ("out", C.c_void_p), ("stats", C.c_void_p),
        ("src0_ld", C.c_int), ("src1_ld", C.c_int), ("out_ld", C.c_int), ("out_coff", C.c_int), ("post_lrelu", C.c_int),
        ("res1", C.c_void_p), ("res1_ld", C.c_int), ("res1_scale", C.c_float),
        ("res2", C.c_void_p), ("res2_ld", C.c_int), ("res2_scale", C.c_float),
        ("post_scale", C.c_void_p), ("post_shift", C.c_void_p), ("post_relu", C.c_int),
        ("io_h16", C.c_int),
        ("bstat_c", C.c_void_p), ("bstat_mean", C.c_void_p), ("bstat_invstd", C.c_void_p), ("bstat_ms", C.c_void_p), ("bstat_mh", C.c_void_p),
        ("stats_clean", C.c_int),
    ]


class HBwd16Args(C.Structure):
    _fields_ = ([("g", C.c_void_p), ("c", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("coef", C.c_void_p), ("k1", C.c_void_p),
                 ("k2", C.c_void_p), ("mask_scale", C.c_void_p), ("mask_shift", C.c_void_p), ("x", C.c_void_p), ("pre_scale", C.c_void_p),
                 ("pre_shift", C.c_void_p), ("pre_relu", C.c_int), ("w", C.c_void_p), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                 ("dx", C.c_void_p), ("dx_b16", C.c_int), ("res", C.c_void_p), ("bstat_c", C.c_void_p), ("bstat_mean", C.c_void_p),
                 ("bstat_invstd", C.c_void_p), ("bstat_ms", C.c_void_p), ("bstat_mh", C.c_void_p), ("stats", C.c_void_p), ("stats_clean", C.c_int),
                 ("dw", C.c_void_p), ("ws", C.c_void_p), ("relu_bits", C.c_void_p)])


class HBlock16Args(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p), ("scale1", C.c_void_p), ("shift1", C.c_void_p),
                ("scale2", C.c_void_p), ("shift2", C.c_void_p), ("out", C.c_void_p), ("out_h16", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int)]


class HWGradArgs(C.Structure):
    _fields_ = [
        ("src0", C.c_void_p), ("c0", C.c_int),
        ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p), ("pre_relu", C.c_int),
        ("src1", C.c_void_p), ("c1", C.c_int),
        ("dy", C.c_void_p), ("cout", C.c_int), ("ksize", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("dw", C.c_void_p), ("ws", C.c_void_p),
        ("src0_ld", C.c_int), ("src1_ld", C.c_int), ("io", C.c_int),
    ]


class BnActArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p),
        ("pooled", C.c_void_p), ("res", C.c_void_p), ("drop", C.c_void_p),
        ("momentum", C.c_float), ("eps", C.c_float), ("B", C.c_int), ("C", C.c_int), ("HW", C.c_int), ("act", C.c_int),
        ("ws", C.c_void_p),
    ]


class MbMidArgs(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("e_pre", "wdw", "gamma0", "beta0", "running_mean0", "running_var0", "mean0", "invstd0",
                                           "gamma1", "beta1", "running_mean1", "running_var1", "mean1", "invstd1", "d_pre", "y", "pooled")]
                + [(n, C.c_float) for n in ("momentum0", "eps0", "momentum1", "eps1")] + [(n, C.c_int) for n in ("B", "C", "H", "W", "K")])


class MbMidBwdArgs(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("dout", "gate", "dpooled", "d_pre", "e_pre", "wdw", "gamma0", "beta0", "mean0", "invstd0",
                                           "gamma1", "beta1", "mean1", "invstd1", "de_pre", "dwdw", "dgamma0", "dbeta0", "dgamma1", "dbeta1")]
                + [(n, C.c_int) for n in ("B", "C", "H", "W", "K")])


class BnActBwdArgs(C.Structure):
    _fields_ = [
        ("dy", C.c_void_p), ("x", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p), ("gate", C.c_void_p), ("dpooled", C.c_void_p), ("drop", C.c_void_p),
        ("dx", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
        ("B", C.c_int), ("C", C.c_int), ("HW", C.c_int), ("act", C.c_int), ("ws", C.c_void_p),
    ]


class ConvW(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p)]


class RRDBNetDesc(C.Structure):
    _fields_ = [
        ("num_in_ch", C.c_int), ("num_block", C.c_int),
        ("conv_first_w", C.c_void_p), ("conv_first_b", C.c_void_p),
        ("rdb", C.POINTER(ConvW)),
        ("conv_body", ConvW), ("conv_up1", ConvW), ("conv_up2", ConvW), ("conv_hr", ConvW),
        ("conv_last", ConvW), ("num_out_ch", C.c_int),
    ]


# every exported symbol of include/srbh.h: name -> (restype, argtypes)
_vp, _i, _sz, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_float
SIGNATURES = {
    "srbh_version": (_i, []),
    "srbh_last_error": (C.c_char_p, []),
    "srbh_path_counters": (_i, [_vp, _i, _i]),
    "srbh_act16_bytes": (_sz, [_i, _i, _i, _i]),
    "srbh_nchw32_to_act16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_act16_to_nchw32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_wpack16_bytes": (_sz, [_i, _i]),
    "srbh_pack_conv3x3_f16": (_i, [_vp, _i, _i, _vp, _vp]),
    "srbh_conv3x3_f16": (_i, [C.POINTER(ConvArgs), _vp]),
    "srbh_conv3x3_x16": (_i, [C.POINTER(ConvArgs), _i, _vp, _i, _i, _vp]),
    "srbh_pack_conv3x3_b16": (_i, [_vp, _i, _i, _vp, _vp]),
    "srbh_nhwc32_to_act16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "srbh_act16_channel_sum": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "srbh_axpby_f32": (_i, [_vp, _f, _vp, _f, _vp, C.c_long, _vp]),
    "srbh_rrdbnet_trunk_train_forward": (_i, [C.POINTER(RRDBNetDesc), _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "srbh_rrdbnet_trunk_train_aux_bytes": (_sz, [_i, _i, _i]),
    "srbh_rrdbnet_trunk_train_forward_persistent": (_i, [C.POINTER(RRDBNetDesc), _vp, _vp, _vp, _sz, _i, _i, _i, _vp, _vp, C.POINTER(C.c_int)]),
    "srbh_rrdbnet_trunk_wgrad_ws_bytes": (_sz, []),
    "srbh_rrdbnet_trunk_train_backward_persistent": (_i, [_i, _vp, _sz, _vp, _sz, C.POINTER(_sz), _vp, _vp, _vp, _vp, C.POINTER(_vp), _vp, _sz, _vp, _vp, _vp, _vp, _i, _i, _i, _vp,
                                                          _vp, C.POINTER(_i)]),
    "srbh_lrelu_bwd_f32": (_i, [_vp, _vp, C.c_float, C.c_long, _vp]),
    "srbh_up2_bwd_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_bilinear2x_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "srbh_pack_conv3x3_many": (_i, [_vp, _i, C.c_long, _vp]),
    "srbh_trunk_wgrad_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "srbh_trunk_wgrad": (_i, [_i, _vp, _sz, _vp, _sz, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "srbh_rrdbnet_trunk_train_backward": (_i, [_i, _vp, _sz, _vp, _sz, C.POINTER(_sz), _vp, _vp, _vp, C.POINTER(_vp), _vp, _sz, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "srbh_act16_wgrad_b16": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "srbh_conv_first_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "srbh_hpack_bytes": (_sz, [_i, _i, _i]),
    "srbh_hpack_conv_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "srbh_bn_stats_bytes": (_sz, [_i]),
    "srbh_hconv_up_supported": (_i, [_i, _i]),
    "srbh_hconv_f32": (_i, [C.POINTER(HConvArgs), _vp]),
    "srbh_hpack_h16_bytes": (_sz, [_i, _i, _i]),
    "srbh_hpack_conv_h16": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "srbh_hpack_conv_h16_many": (_i, [_vp, _i, C.c_long, _vp]),
    "srbh_hconv_h16": (_i, [C.POINTER(HConvArgs), _i, _vp]),
    "srbh_hconv_entry_h16": (_i, [C.POINTER(HConvArgs), C.POINTER(HConvArgs), _i, _vp]),
    "srbh_bn_finalize": (_i, [_vp, _i, C.c_double, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "srbh_bn_finalize_clear": (_i, [_vp, _i, C.c_double, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "srbh_bn_eval_scale_shift": (_i, [_i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "srbh_bn_add_relu": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_long, _i, _vp]),
    "srbh_bn_add_relu_io": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_long, _i, _i, _vp]),
    "srbh_relu_bits_bytes": (_sz, [C.c_long, _i]),
    "srbh_bn_add_relu_bits": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_long, _i, _i, _vp]),
    "srbh_aggregate": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_nearest2x_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_nchw_to_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_hconv_wgrad_f32": (_i, [C.POINTER(HWGradArgs), _vp]),
    "srbh_hconv_wgrad_b16": (_i, [C.POINTER(HWGradArgs), _vp]),
    "srbh_hconv_wgrad_entry_b16": (_i, [C.POINTER(HWGradArgs), C.POINTER(HWGradArgs), _vp]),
    "srbh_adam_chunk": (_i, []),
    "srbh_adam_step": (_i, [_vp, _vp, _i, C.c_double, C.c_double, C.c_double, _vp]),
    "srbh_hbwd16_supported": (_i, [_i, _i]),
    "srbh_hbwd16": (_i, [C.POINTER(HBwd16Args), _vp]),
    "srbh_hblock16_supported": (_i, [_i, _i]),
    "srbh_hblock16_eval": (_i, [C.POINTER(HBlock16Args), _vp]),
    "srbh_hwgrad_defer": (_i, [_i]),
    "srbh_hwgrad_flush": (_i, [_vp]),
    "srbh_relu_mask_mul": (_i, [_vp, _vp, _vp, C.c_long, _vp]),
    "srbh_add_inplace": (_i, [_vp, _vp, C.c_long, _vp]),
    "srbh_bn_bwd_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_long, _i, _vp, _vp]),
    "srbh_bn_bwd_reduce_relu": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_long, _i, _vp, _vp]),
    "srbh_bn_bwd_finalize": (_i, [_vp, _i, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "srbh_bn_bwd_finalize_clear": (_i, [_vp, _i, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "srbh_bn_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_long, _i, _vp]),
    "srbh_bn_bwd_reduce_io": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_long, _i, _vp, _i, _vp]),
    "srbh_bn_bwd_apply_io": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_long, _i, _i, _vp]),
    "srbh_hwgrad_ws_bytes": (_sz, [_i, _i, _i]),
    "srbh_ps2_inverse": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_affine_act_nchw": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_stem_conv_eval_supported": (_i, [_i, _i, _i]),
    "srbh_stem_conv_eval": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "srbh_affine_act_add_nchw": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_affine_act_pool_nchw": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_se_hidden": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "srbh_se_gate_scale": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_se_gate": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "srbh_dwconv_eval_supported": (_i, [_i] * 10),
    "srbh_dwconv_eval_fwd": (_i, [_vp] * 8 + [_i] * 10 + [_vp]),
    "srbh_bn_act_train_supported": (_i, [_i, _i, _i]),
    "srbh_bn_act_train_ws_bytes": (_sz, [_i, _i, _i]),
    "srbh_bn_act_train_fwd": (_i, [C.POINTER(BnActArgs), _vp]),
    "srbh_bn_act_train_bwd": (_i, [C.POINTER(BnActBwdArgs), _vp]),
    "srbh_se_train_fwd": (_i, [_vp] * 9 + [_i] * 4 + [_vp]),
    "srbh_se_train_bwd": (_i, [_vp] * 18 + [_i] * 5 + [_vp]),
    "srbh_se_train_bwd_ws_floats": (_sz, [_i, _i, _i]),
    "srbh_up2_cat_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "srbh_up2_cat_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "srbh_mbconv_mid_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "srbh_mbconv_mid_fwd": (_i, [C.POINTER(MbMidArgs), _vp]),
    "srbh_mbconv_mid_bwd": (_i, [C.POINTER(MbMidBwdArgs), _vp]),
    "srbh_dconv_pack_many": (_i, [_vp, _i, _vp]),
    "srbh_dconv_supported": (_i, [_i, _i, _i, _i, _i]),
    "srbh_dconv_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "srbh_dconv_fwd_epi": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "srbh_dconv_wgrad_ws_floats": (_sz, [_i, _i, _i, _i, _i]),
    "srbh_dconv_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "srbh_pwconv_supported": (_i, [_i, _i, _i, _i]),
    "srbh_pwconv_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_pwconv_fwd_wt": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_pwconv_fwd_epi": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "srbh_transpose_many": (_i, [_vp, _i, _vp]),
    "srbh_pwconv_bwd_data": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_pwconv_bwd_data_res": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_pwconv_bwd_weight_ws_floats": (_sz, [_i, _i, _i, _i]),
    "srbh_pwconv_bwd_weight": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srbh_dwconv_fwd": (_i, [_vp, _vp, _vp] + [_i] * 10 + [_vp]),
    "srbh_dwconv_bwd_data": (_i, [_vp, _vp, _vp] + [_i] * 10 + [_vp]),
    "srbh_dwconv_bwd_weight_splits": (_i, [_i, _i]),
    "srbh_dwconv_bwd_weight": (_i, [_vp, _vp, _vp, _vp] + [_i] * 10 + [_vp]),
    "srbh_mosaic_accumulate": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "srbh_mosaic_finalize": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "srbh_label_prep": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "srbh_normalize_clamp": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _f, _f, _i, _vp]),
    "srbh_rrdbnet_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "srbh_rrdbnet_last_status": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "srbh_ptail_wgs_cap": (_i, [_i]),
    "srbh_wmse_sum": (_i, [_vp, _vp, _vp, C.c_long, _vp, _vp]),
    "srbh_wmse_grad": (_i, [_vp, _vp, _vp, C.c_long, _vp, _vp, _vp]),
    "srbh_cedice_sums": (_i, [_vp, _i, _i, C.c_long, C.c_long, C.c_long, C.c_long, _vp, _vp, _vp, _vp]),
    "srbh_cedice_grad": (_i, [_vp, _i, _i, C.c_long, C.c_long, C.c_long, C.c_long, _vp, _vp, _vp, _vp, _vp]),
    "srbh_height_metric_sums": (_i, [_vp, _vp, _vp, C.c_long, _i, _vp, _vp]),
    "srbh_confusion_add": (_i, [_vp, _vp, C.c_long, _i, _vp, _vp, _vp]),
    "srbh_trunk_timing": (_i, [_i]),
    "srbh_trunk_last_ms": (_i, [C.POINTER(C.c_float)]),
    "srbh_trunk_kernel_name": (C.c_char_p, []),
    "srbh_rrdbnet_forward": (_i, [C.POINTER(RRDBNetDesc), _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
}

_lib = None
_lock = threading.Lock()


def lib() -> C.CDLL:
    """Load libsrbh.so (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"libsrbh.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback for the HIP hot path.")
                l = C.CDLL(_DEV_LIB or LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(l, name)  # AttributeError here == header/library mismatch
                    fn.restype = res
                    fn.argtypes = args
                _lib = l
    return _lib


_HOST_SPIN_US = float(os.environ.get("SRBH_HOST_SPIN_US", "0"))      # developer probe (tools/r05_14.sh): burn host time per libsrbh call -- is a workload host-bound?


def check(rc: int, what: str = "") -> None:
    if _HOST_SPIN_US:
        import time
        t_end = time.perf_counter() + _HOST_SPIN_US * 1e-6
        while time.perf_counter() < t_end:
            pass
    if rc != 0:
        msg = lib().srbh_last_error().decode(errors="replace")
        raise RuntimeError(f"libsrbh {what} failed (rc={rc}): {msg}")


_RAW_STREAM = None


def stream_ptr():
    """the current HIP stream of the current device as a void* for the C-ABI.  torch.cuda.current_stream() builds a Stream object through
    ~10 Python frames (9 us: 6.6 ms of host time per training step at ~700 libsrbh calls -- tools/host_bwd_profile.py); the raw getters
    underneath it take 0.3 us.  Falls back to the public API where the private ones are missing."""
    global _RAW_STREAM
    if _RAW_STREAM is None:
        import torch
        get_raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        get_dev = getattr(torch._C, "_cuda_getDevice", None)
        if get_raw is not None and get_dev is not None:
            _RAW_STREAM = lambda: get_raw(get_dev())                     # noqa: E731
        else:
            _RAW_STREAM = lambda: torch.cuda.current_stream().cuda_stream    # noqa: E731
    return C.c_void_p(_RAW_STREAM())
