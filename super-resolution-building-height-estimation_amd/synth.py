"""Deterministic synthetic weights / inputs (seeded): used by bench.py, the harness and -- through oracle/synth.py -- the tests.

No checkpoint ships with the reference, so every parity test runs on seeded
synthetic weights.  Each tensor gets its own ``torch.Generator`` seeded from crc32(key)^seed, so
fixtures never have to store weights and key order does not matter.

``mode='init'`` follows the reference initialisers (RDB convs: kaiming-normal x0.1, zero bias --
SR/rrdbnet_arch.py:20-48,134; everything else torch defaults).  ``mode='stress'`` additionally
randomises biases and BatchNorm affine/running statistics so bias / BN code paths cannot hide.
"""
from __future__ import annotations

import math
import zlib

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def _conv(sd, name, cout, cin, k, seed, bias=True, kaiming_scale=None, mode="init"):
    fan_in = cin * k * k
    g = _gen(name + ".weight", seed)
    if kaiming_scale is not None:
        w = torch.randn(cout, cin, k, k, generator=g) * (math.sqrt(2.0 / fan_in) * kaiming_scale)
    else:
        bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform(a=sqrt(5))
        w = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound
    sd[name + ".weight"] = w
    if bias:
        gb = _gen(name + ".bias", seed)
        if kaiming_scale is not None and mode == "init":
            b = torch.zeros(cout)
        else:
            b = (torch.rand(cout, generator=gb) * 2 - 1) * (1.0 / math.sqrt(fan_in))
        sd[name + ".bias"] = b


def _bn(sd, name, c, seed, mode):
    if mode == "stress":
        g = _gen(name, seed)
        sd[name + ".weight"] = 0.5 + torch.rand(c, generator=g)
        sd[name + ".bias"] = (torch.rand(c, generator=g) - 0.5) * 0.4
        sd[name + ".running_mean"] = (torch.rand(c, generator=g) - 0.5) * 0.2
        sd[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)
    else:
        sd[name + ".weight"] = torch.ones(c)
        sd[name + ".bias"] = torch.zeros(c)
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.ones(c)
    sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def rrdbnet_state_dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32,
                       seed=1337, mode="init"):
    """Keys/shapes of RRDBNet.state_dict() (SR/rrdbnet_arch.py:190-206)."""
    if scale == 2:
        num_in_ch *= 4
    elif scale == 1:
        num_in_ch *= 16
    sd = {}
    _conv(sd, "conv_first", num_feat, num_in_ch, 3, seed, mode=mode)
    for i in range(num_block):
        for r in (1, 2, 3):
            p = f"body.{i}.rdb{r}."
            for k in range(1, 5):
                _conv(sd, f"{p}conv{k}", num_grow_ch, num_feat + (k - 1) * num_grow_ch, 3, seed,
                      kaiming_scale=0.1, mode=mode)
            _conv(sd, f"{p}conv5", num_feat, num_feat + 4 * num_grow_ch, 3, seed, kaiming_scale=0.1, mode=mode)
    for n in ("conv_body", "conv_up1", "conv_up2", "conv_hr"):
        _conv(sd, n, num_feat, num_feat, 3, seed, mode=mode)
    _conv(sd, "conv_last", num_out_ch, num_feat, 3, seed, mode=mode)
    return sd


def basicblock_state_dict(sd, p, inplanes, planes, seed, mode):
    """BasicBlock keys (SR/HRfuse.py:129-140)."""
    _conv(sd, p + "conv1", planes, inplanes, 3, seed, bias=False, mode=mode)
    _bn(sd, p + "bn1", planes, seed, mode)
    _conv(sd, p + "conv2", planes, planes, 3, seed, bias=False, mode=mode)
    _bn(sd, p + "bn2", planes, seed, mode)
    if inplanes != planes:
        _conv(sd, p + "downsample.0", planes, inplanes, 1, seed, bias=False, mode=mode)
        _bn(sd, p + "downsample.1", planes, seed, mode)


def hrfeature_state_dict(in_chans, mid_chans=64, out_chans=64, seed=1337, mode="init", prefix=""):
    sd = {}
    basicblock_state_dict(sd, prefix + "0.", in_chans, mid_chans, seed, mode)
    basicblock_state_dict(sd, prefix + "1.", mid_chans, mid_chans, seed, mode)
    basicblock_state_dict(sd, prefix + "2.", mid_chans, out_chans, seed, mode)
    return sd


def hrfuse_residual_state_dict(hr_chans=16, lr_chans=16, mid_chans=16, out_chans=3, upscale=4, seed=1337,
                               mode="init", prefix=""):
    sd = {}
    idx, s = 0, upscale
    while s > 1:
        _conv(sd, f"{prefix}upsampler.{idx}", 4 * lr_chans, lr_chans, 3, seed, mode=mode)
        idx += 2
        s //= 2
    basicblock_state_dict(sd, prefix + "fuse.0.", hr_chans + lr_chans, mid_chans, seed, mode)
    basicblock_state_dict(sd, prefix + "fuse.1.", mid_chans, mid_chans, seed, mode)
    basicblock_state_dict(sd, prefix + "fuse.2.", mid_chans, mid_chans, seed, mode)
    _conv(sd, prefix + "conv_last", out_chans, mid_chans, 3, seed, mode=mode)
    return sd


def tiles(batch, chans=8, size=64, seed=1337, kind="train"):
    """Synthetic Sentinel tiles: uniform[0,1) like the clipped training tiles
    (BH_loader.py:361-369) or N(0.35,0.25) unclipped like the grid loader (BH_loader.py:984-986)."""
    g = torch.Generator()
    g.manual_seed(seed)
    if kind == "train":
        return torch.rand(batch, chans, size, size, generator=g)
    return torch.randn(batch, chans, size, size, generator=g) * 0.25 + 0.35


def clone_sd(sd):
    return {k: v.clone() for k, v in sd.items()}
