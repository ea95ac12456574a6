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
        self._rows = [self.H, 0]     # [first, last+1) mosaic rows this instance has written (for reduce_to_)

    def add(self, ypred, build_pred, posall):
        """ypred (N,1,h,w), build_pred (N,C,h,w) raw logits, posall (N,4) = (xoff,yoff,xcount,ycount) in LR cells
        (multiplied by 4 here, as predict...py:182 does)."""
        if not (ypred.is_cuda and build_pred.is_cuda):
            raise RuntimeError("Mosaic.add (libsrbh): device tensors only (no CPU fallback)")
        n, _, th, tw = ypred.shape
        hv = ypred.detach().float().contiguous()
        bl = H.to_nhwc(build_pred.detach().float())
        pos_h = torch.as_tensor(posall, dtype=torch.int32).reshape(n, 4) * 4
        if n:
            self._rows[0] = min(self._rows[0], max(0, int(pos_h[:, 1].min())))
            self._rows[1] = max(self._rows[1], min(self.H, int((pos_h[:, 1] + pos_h[:, 3]).max())))
        pos = pos_h.to(ypred.device).contiguous()
        _lib.check(_lib.lib().srbh_mosaic_accumulate(hv.data_ptr(), bl.data_ptr(), self.C, n, th, tw, pos.data_ptr(),
                                                     self.res_height.data_ptr(), self.res_build.data_ptr(),
                                                     self.res_weight.data_ptr(), self.H, self.W, _lib.stream_ptr()),
                   "mosaic_accumulate")

    def merge_(self, other):
        """integer sums are additive: the result does not depend on how tiles were sharded over ranks."""
        self.res_height += other.res_height
        self.res_build += other.res_build
        self.res_weight += other.res_weight
        if other._rows[1] > other._rows[0]:      # the written band is now the union (reduce_to_ ships exactly this band)
            self._rows = [min(self._rows[0], other._rows[0]), max(self._rows[1], other._rows[1])]
        return self

    def all_reduce_(self, dist):
        for t in (self.res_height, self.res_build, self.res_weight):
            dist.all_reduce(t)
        self._rows = [0, self.H]                 # every rank now holds every rank's rows
        return self

    def reduce_to_(self, dist, dst=0):
        """Sum the ranks' mosaics into rank `dst` moving only the ROW BAND each rank has written (contiguous shards of a
        row-major grid touch contiguous bands: an 8-way shard of a 27k x 27k city gathers 27 GB / 8 per rank instead of
        all-reducing 27 GB on every rank).  Afterwards only `dst` holds the city."""
        world, rank = dist.get_world_size(), dist.get_rank()
        dev = self.res_height.device
        y0, y1 = (self._rows[0], self._rows[1]) if self._rows[1] > self._rows[0] else (0, 0)
        mine = torch.tensor([y0, y1], dtype=torch.int64, device=dev)
        ranges = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(ranges, mine)
        ranges = [(int(r[0]), int(r[1])) for r in ranges]
        on_host = dist.get_backend() == "gloo"          # (gloo moves host tensors; RCCL moves device tensors)
        # point-to-point, one band at a time, each exactly as tall as what that rank wrote: `dst` holds ONE receive
        # buffer (the tallest band) next to its mosaic instead of world x padded bands (which doubled its memory for a
        # 27k x 27k city), and the senders' copies overlap with dst's accumulation of the previous band
        if rank != dst:
            n = y1 - y0
            if n:
                pack = torch.empty((self.C + 2, n, self.W), dtype=torch.int32, device=dev)
                pack[0] = self.res_height[y0:y1]
                pack[1] = self.res_weight[y0:y1]
                pack[2:] = self.res_build[:, y0:y1]
                dist.send(pack.cpu() if on_host else pack, dst=dst)
            return self
        for r, (a, b) in enumerate(ranges):
            if r == dst or b <= a:
                continue
            band = torch.empty((self.C + 2, b - a, self.W), dtype=torch.int32, device="cpu" if on_host else dev)
            dist.recv(band, src=r)
            band = band.to(dev) if on_host else band
            self.res_height[a:b] += band[0]
            self.res_weight[a:b] += band[1]
            self.res_build[:, a:b] += band[2:]
            del band
        self._rows = [min([a for a, b in ranges if b > a] or [self.H]), max([b for a, b in ranges] or [0])]
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
