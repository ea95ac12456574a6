"""Device-resident integer mosaics for the urban-centre predict path (reference
predict_realesanet_feature_globe.py:156-204): quantise the two model outputs, scatter-add the tiles,
finalise with argmax / rounded division.  Exact integer semantics => shards merge by plain addition."""
from __future__ import annotations

import torch

from . import _lib
from . import hrfuse as H

__all__ = ["Mosaic"]


class Mosaic:
    def __init__(self, height, width, chans_build, device):
        self.H, self.W, self.C = int(height), int(width), int(chans_build)
        self.res_height = torch.zeros((self.H, self.W), dtype=torch.int32, device=device)     # uint32 bit patterns
        self.res_build = torch.zeros((self.C, self.H, self.W), dtype=torch.int32, device=device)
        self.res_weight = torch.zeros((self.H, self.W), dtype=torch.int32, device=device)

    def add(self, ypred, build_pred, posall):
        """ypred (N,1,h,w), build_pred (N,C,h,w) raw logits, posall (N,4) = (xoff,yoff,xcount,ycount) in LR cells
        (multiplied by 4 here, as predict...py:182 does)."""
        if not (ypred.is_cuda and build_pred.is_cuda):
            raise RuntimeError("Mosaic.add (libsrbh): device tensors only (no CPU fallback)")
        n, _, th, tw = ypred.shape
        hv = ypred.detach().float().contiguous()
        bl = H.to_nhwc(build_pred.detach().float())
        pos = (torch.as_tensor(posall, dtype=torch.int32).reshape(n, 4) * 4).to(ypred.device).contiguous()
        _lib.check(_lib.lib().srbh_mosaic_accumulate(hv.data_ptr(), bl.data_ptr(), self.C, n, th, tw, pos.data_ptr(),
                                                     self.res_height.data_ptr(), self.res_build.data_ptr(),
                                                     self.res_weight.data_ptr(), self.H, self.W, _lib.stream_ptr()),
                   "mosaic_accumulate")

    def merge_(self, other):
        """integer sums are additive: the result does not depend on how tiles were sharded over ranks."""
        self.res_height += other.res_height
        self.res_build += other.res_build
        self.res_weight += other.res_weight
        return self

    def all_reduce_(self, dist):
        for t in (self.res_height, self.res_build, self.res_weight):
            dist.all_reduce(t)
        return self

    def finalize(self):
        """-> (height uint16 as int32 tensor view-safe: torch.uint16, build class uint8), shapes (H,W)."""
        dev = self.res_height.device
        hout = torch.empty((self.H, self.W), dtype=torch.uint16, device=dev)
        bout = torch.empty((self.H, self.W), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().srbh_mosaic_finalize(self.res_height.data_ptr(), self.res_build.data_ptr(),
                                                   self.res_weight.data_ptr(), self.C, self.H, self.W, hout.data_ptr(),
                                                   bout.data_ptr(), _lib.stream_ptr()), "mosaic_finalize")
        return hout, bout
