"""Thin test-side wrappers that drive individual libsrbh entry points through the C ABI."""
import ctypes as C

import torch
import torch.nn.functional as F

from srbh_amd import _lib


def act16_from_nchw(x, chunks_total=None):
    """(B,C,H,W) fp32 cuda -> zero-bordered ACT16 buffer (uint8 tensor)."""
    L = _lib.lib()
    B, Cc, H, W = x.shape
    ch = chunks_total or (Cc + 31) // 32
    buf = torch.zeros(L.srbh_act16_bytes(B, ch * 32, H, W), dtype=torch.uint8, device=x.device)
    _lib.check(L.srbh_nchw32_to_act16(x.contiguous().data_ptr(), buf.data_ptr(), B, Cc, H, W, _lib.stream_ptr()))
    return buf


def act16_alloc(B, chunks, H, W, device):
    return torch.zeros(_lib.lib().srbh_act16_bytes(B, chunks * 32, H, W), dtype=torch.uint8, device=device)


def act16_to_nchw(buf, B, Cc, H, W):
    out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=buf.device)
    _lib.check(_lib.lib().srbh_act16_to_nchw32(buf.data_ptr(), out.data_ptr(), B, Cc, H, W, _lib.stream_ptr()))
    return out


def pack_w(w):
    L = _lib.lib()
    cout, cin = w.shape[:2]
    buf = torch.zeros(L.srbh_wpack16_bytes(cout, cin), dtype=torch.uint8, device=w.device)
    _lib.check(L.srbh_pack_conv3x3_f16(w.contiguous().data_ptr(), cout, cin, buf.data_ptr(), _lib.stream_ptr()))
    return buf


def conv_args(**kw):
    a = _lib.ConvArgs()
    for k, v in kw.items():
        setattr(a, "in_" if k == "in" else k, v)
    return a


def run_conv(a):
    _lib.check(_lib.lib().srbh_conv3x3_f16(C.byref(a), _lib.stream_ptr()), "conv3x3_f16")


def h16(t):
    """round to fp16 and back (what the MFMA operands see)."""
    return t.half().float()


def ref_conv(x, w, b, ups=False, rounded=True):
    """CPU fp32 conv of (optionally fp16-rounded) operands, fp32 accumulate (double for stability)."""
    x = x.double() if not rounded else h16(x).double()
    w = w.double() if not rounded else h16(w).double()
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(x, w, None if b is None else b.double(), 1, 1).float()
