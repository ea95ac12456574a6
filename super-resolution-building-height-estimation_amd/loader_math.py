"""Loader-side tensor math on the device (reference BH_loader.py:30-61,326-329,361-392).

``hierweight*`` are the (tiny, host-side) class-weight formulas; ``LabelPrep`` / ``normalize_tiles`` are libsrbh
kernels that turn a batch of raw uint8 height labels / raw band values into the tensors the training step consumes."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

__all__ = ["hierweight", "hierweight_simple", "hierweight_equal", "LabelPrep", "normalize_tiles"]


def _class_freq(stats, hir):
    stats = np.asarray(stats, dtype=np.float64)
    stats = stats / stats.sum()
    return np.array([stats[hir[i]:hir[i + 1]].sum() for i in range(len(hir) - 1)])


def _rescale(w):
    w = w / w.sum()
    return len(w) / np.sum(w) * w


def hierweight(stats, hir):
    """inverse square-root class frequency, rescaled so that the weights sum to the class count (BH_loader.py:30-41)."""
    return _rescale(1.0 / np.sqrt(_class_freq(stats, hir)))


def hierweight_simple(stats, hir):
    """plain inverse frequency (BH_loader.py:44-55)."""
    return _rescale(1.0 / _class_freq(stats, hir))


def hierweight_equal(stats, hir):
    return np.ones((len(hir) - 1,))


class LabelPrep:
    """buildhir LUT + class weights on the device (BH_loader.py:326-329,373-392)."""

    def __init__(self, hir, heightweight, device):
        lut = np.zeros((256,), dtype=np.uint8)
        for i in range(len(hir) - 1):
            lut[hir[i]:hir[i + 1]] = i
        self.lut = torch.from_numpy(lut).to(device)
        self.cw = torch.as_tensor(np.asarray(heightweight), dtype=torch.float32, device=device)

    def __call__(self, height_u8):
        """height_u8: (B,H,W) uint8 device tensor -> (height [B,H,W] f32, height_aggre [B,H/4,W/4], build int64,
        weight, weight_aggre)"""
        if not (height_u8.is_cuda and height_u8.dtype == torch.uint8 and height_u8.dim() == 3):
            raise RuntimeError("LabelPrep (libsrbh): expects a (B,H,W) uint8 device tensor")
        h = height_u8.contiguous()
        B, Hh, Ww = h.shape
        dev = h.device
        build = torch.empty((B, Hh, Ww), dtype=torch.int64, device=dev)
        hf = torch.empty((B, Hh, Ww), dtype=torch.float32, device=dev)
        wt = torch.empty((B, Hh, Ww), dtype=torch.float32, device=dev)
        ha = torch.empty((B, Hh // 4, Ww // 4), dtype=torch.float32, device=dev)
        wa = torch.empty((B, Hh // 4, Ww // 4), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().srbh_label_prep(h.data_ptr(), B, Hh, Ww, self.lut.data_ptr(), self.cw.data_ptr(),
                                              build.data_ptr(), hf.data_ptr(), wt.data_ptr(), ha.data_ptr(), wa.data_ptr(),
                                              _lib.stream_ptr()), "label_prep")
        return hf, ha, build, wt, wa


def normalize_tiles(img, mins, maxs, datarange=(0, 1)):
    """(img - min) / (max - min) per band then clip (BH_loader.py:361-369; `normmethod='minmax'`, :304-306)."""
    if not img.is_cuda:
        raise RuntimeError("normalize_tiles (libsrbh): device tensors only")
    x = img.detach().float().contiguous()
    B, Cc, Hh, Ww = x.shape
    mn = torch.as_tensor(np.asarray(mins), dtype=torch.float32, device=x.device)
    rg = torch.as_tensor(np.asarray(maxs) - np.asarray(mins), dtype=torch.float32, device=x.device)
    out = torch.empty_like(x)
    clamp = isinstance(datarange, tuple)
    lo, hi = (datarange if clamp else (0.0, 0.0))
    _lib.check(_lib.lib().srbh_normalize_clamp(x.data_ptr(), out.data_ptr(), B, Cc, Hh, Ww, mn.data_ptr(), rg.data_ptr(),
                                               float(lo), float(hi), int(clamp), _lib.stream_ptr()), "normalize_clamp")
    return out
