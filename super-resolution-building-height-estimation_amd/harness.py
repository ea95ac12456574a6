"""Callers of the hot path, reproduced just far enough to drive and measure it.

The reference's train.py / predict script are out of scope; this module restates their *call sequence*
around the two networks so that a full training step (BASELINE config 3), its data-parallel form (config 4: one
process per GPU, gradients averaged by one bucketed RCCL all-reduce per step) and tile-sharded inference (config 5) can
be exercised.  Losses follow losses_pytorch/selfloss.py:6-17,81-91,145-168 in plain torch ops (they are the "next"
row 8f-3, not yet kernels).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

HIR = (0, 3, 12, 21, 30, 60, 90, 256)                       # train.py:55
# hierweight(bh_stats_globe, HIR) as probed on the reference data; synthetic labels reuse it
CLASS_WEIGHT = (0.08878965, 0.272375, 0.32563883, 0.74654097, 0.99655239, 1.62749907, 2.94260409)


from .losses import MSE_adapt_weight, CE_DICE_adapt_weight   # noqa: E402  (libsrbh reductions, losses.py)


def synthetic_batch(batch, seed, device, aggregate=None):
    """One training batch shaped like myImageFloder_S12_globe's output (BH_loader.py:338-400): 8-channel 64x64 tile in
    [0,1), 256x256 height label with ~82% zeros in blocky 8x8 patches, hierarchy class map, per-pixel class weights and
    their 4x4 aggregates.  ``aggregate`` defaults to the libsrbh aggregate_torch kernel on `device`."""
    g = torch.Generator()
    g.manual_seed(seed)
    lr = torch.rand(batch, 8, 64, 64, generator=g)
    coarse = torch.rand(batch, 1, 32, 32, generator=g)
    hval = torch.rand(batch, 1, 32, 32, generator=g) ** 3 * 120.0
    height = torch.where(coarse > 0.82, hval, torch.zeros_like(hval))
    height = F.interpolate(height, scale_factor=8, mode="nearest").round()
    edges = torch.tensor(HIR[1:-1], dtype=torch.float32)
    build = torch.bucketize(height[:, 0], edges, right=True)
    build = torch.where(height[:, 0] <= 0, torch.zeros_like(build), build).long().clamp_(0, 6)
    weight = torch.tensor(CLASS_WEIGHT)[build]
    lr, height, build, weight = lr.to(device), height.to(device), build.to(device), weight.to(device)
    if aggregate is None:
        from .aggregate import aggregate_torch as aggregate
    height_aggre = aggregate(height, 0.25).reshape(batch, 64, 64)
    weight_aggre = aggregate(weight[:, None].contiguous(), 0.25).reshape(batch, 64, 64)
    return lr, height[:, 0], height_aggre, build, weight, weight_aggre


def shard_range(n_items, rank, world):
    """contiguous, balanced shard of range(n_items) for `rank` (tiles are independent: no data-path collective)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_grads(params, world, dist=None, bucket_bytes=64 << 20):
    """Average gradients over ranks with few large all-reduces (RCCL over xGMI is per-link bound: prefer big buckets).
    Parameters without a gradient (e.g. the unused encoder._conv_head) are skipped on every rank alike."""
    if world <= 1:
        return 0
    if dist is None:
        import torch.distributed as dist
    grads = [p.grad for p in params if p.grad is not None]
    buckets, cur, size = [], [], 0
    for gr in grads:
        cur.append(gr)
        size += gr.numel() * gr.element_size()
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    for b in buckets:
        flat = torch.cat([gr.reshape(-1) for gr in b])
        dist.all_reduce(flat)
        flat.div_(world)
        off = 0
        for gr in b:
            n = gr.numel()
            gr.copy_(flat[off:off + n].view_as(gr))
            off += n
    return len(buckets)


class TrainStep:
    """train.py:133-179,243-256: frozen RRDBNet feature extractor + trainable SRRegress_Cls_feature, three
    uncertainty-weighted losses, Adam(lr 1e-3, wd 1e-4) with the log_vars as an extra param group."""

    def __init__(self, net_hr, net, device, world=1, lr=1e-3, sync_bn=False):
        if sync_bn and world > 1:
            # global-batch BatchNorm statistics (what the reference computes on one device): libsrbh BatchNorms all-reduce
            # their partial sums (hrfuse.set_bn_sync), the stock-op encoder / decoders become torch SyncBatchNorm
            from . import hrfuse
            hrfuse.set_bn_sync(world)
            for name in ("encoder", "decoder1", "decoder2"):
                if hasattr(net, name):
                    setattr(net, name, nn.SyncBatchNorm.convert_sync_batchnorm(getattr(net, name)))
        self.net_hr, self.net, self.world = net_hr.eval(), net.train(), world
        for p in self.net_hr.parameters():
            p.requires_grad_(False)
        self.criterion = [MSE_adapt_weight(device=device), MSE_adapt_weight(device=device),
                          CE_DICE_adapt_weight(device=device)]
        self.optimizer = torch.optim.Adam(net.parameters(), lr=lr, weight_decay=1e-4)
        self.optimizer.add_param_group({"params": [c.log_var for c in self.criterion], "lr": lr})
        self.rgbseq = [0, 1, 2]

    def params(self):
        return [p for g in self.optimizer.param_groups for p in g["params"]]

    def __call__(self, batch):
        lr, height, height_aggre, build, weight, weight_aggre = batch
        with torch.no_grad():
            hr_fea = self.net_hr.forward_feature(lr[:, self.rgbseq])
        height_pred, build_pred, height_pred_aggre = self.net(lr, hr_fea)
        loss = (self.criterion[0](height_pred.squeeze(1), height, weight)
                + self.criterion[1](height_pred_aggre.squeeze(1), height_aggre, weight_aggre)
                + self.criterion[2](build_pred, build, weight))
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        allreduce_grads(self.params(), self.world)
        self.optimizer.step()
        return loss.detach(), height_pred.detach()


@torch.no_grad()
def predict_tiles(net_hr, model, tiles, posall, mosaic, batch=32, rank=0, world=1, pad_to=None):
    """predict_whole_image_grid's inner loop (predict_realesanet_feature_globe.py:167-185) for this rank's shard of a
    city's grid cells: RRDBNet feature extraction -> height / building heads -> quantise + integer mosaic on the
    device.  ``tiles`` (N,8,64,64) fp32, ``posall`` (N,4) LR-cell windows.  Merge shards with ``mosaic.all_reduce_``
    (or ``merge_``) before ``mosaic.finalize()``; integer sums make the result independent of the sharding.
    (Producing batch i+1's features on a second stream while batch i's heads run measured no gain: the persistent trunk
    kernel owns every CU while it runs, the launches simply serialise.)"""
    model.eval()
    net_hr.eval()
    lo, hi = shard_range(tiles.shape[0], rank, world)
    dev = mosaic.res_height.device
    for s in range(lo, hi, batch):
        e = min(s + batch, hi)
        x = tiles[s:e].to(dev, non_blocking=True)
        k = e - s
        if k < batch:
            # ragged tail: run a padded batch (eval mode: tiles are independent) instead of a new tensor shape, for which the
            # stock-op encoder would first search / compile kernels (0.5 s per new shape, more than a small city's work).
            # pad_to = granularity of the padded sizes (None: always the full batch; e.g. 32 with batch 128 -> four shapes,
            # which the caller should have warmed up once)
            q = batch if not pad_to else min(batch, (k + pad_to - 1) // pad_to * pad_to)
            if q > k:
                x = torch.cat([x, x.new_zeros((q - k,) + tuple(x.shape[1:]))], 0)
        hr_fea = net_hr.forward_feature(x[:, :3])
        out = model(x, hr_fea)
        mosaic.add(out[0][:k], out[1][:k], posall[s:e])
    return hi - lo
