"""CPU oracle for the SR building-height hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a *functional* fp32 restatement (plain torch CPU ops over a flat
``state_dict``) of the reference modules on the hot path.  It is the checker the
HIP kernels are compared against; it is never imported by the product package
(``super-resolution-building-height-estimation_amd``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity status: PINNED.  ``tools/make_golden.py`` (run in the build container, where
``/root/reference`` is importable) checks every function here against the imported
reference modules (state_dict fed to both, outputs equal to <=1e-6 rel) and writes the
fixtures under ``tests/golden/`` that ``tests/test_oracle_golden.py`` replays on any box.

Reference citations are ``/root/reference``-relative ``file:line``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2  # SR/rrdbnet_arch.py:131,206
RES_SCALE = 0.2    # SR/rrdbnet_arch.py:143,167
BN_EPS = 1e-5      # nn.BatchNorm2d default used by SR/HRfuse.py:124
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------- index maps
def nearest2x_index(n_out: int) -> np.ndarray:
    """Source index of ``F.interpolate(scale_factor=2, mode='nearest')`` along one axis
    (SR/rrdbnet_arch.py:219-220,236-237): out[i] = in[i // 2]."""
    return np.arange(n_out, dtype=np.int64) // 2


def nearest2x(x: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) -> (B,C,2H,2W) by the pure index map above (bit-exact copy)."""
    h, w = x.shape[-2:]
    iy = torch.from_numpy(nearest2x_index(2 * h))
    ix = torch.from_numpy(nearest2x_index(2 * w))
    return x[..., iy[:, None], ix[None, :]]


def pixelshuffle_index(c_out: int, r: int):
    """PixelShuffle(r) source map (SR/HRfuse.py:23,33): out[n,c,r*h+i,r*w+j] = in[n,c*r*r+i*r+j,h,w].
    Returns a function (c, Y, X) -> (c_in, h, w)."""
    def src(c, y, x):
        return c * r * r + (y % r) * r + (x % r), y // r, x // r
    return src


def pixel_shuffle(x: torch.Tensor, r: int) -> torch.Tensor:
    """Bit-exact PixelShuffle written as an explicit gather (no nn.PixelShuffle)."""
    b, c, h, w = x.shape
    co = c // (r * r)
    Y = torch.arange(h * r)
    X = torch.arange(w * r)
    cc = torch.arange(co)
    cin = cc[:, None, None] * r * r + (Y[None, :, None] % r) * r + (X[None, None, :] % r)
    hh = (Y // r)[None, :, None].expand(co, h * r, w * r)
    ww = (X // r)[None, None, :].expand(co, h * r, w * r)
    return x[:, cin, hh, ww]


def pixel_unshuffle(x: torch.Tensor, scale: int) -> torch.Tensor:
    """SR/rrdbnet_arch.py:94-110."""
    b, c, hh, hw = x.shape
    if hh % scale or hw % scale:
        raise AssertionError("pixel_unshuffle: spatial size not divisible by scale")
    h, w = hh // scale, hw // scale
    v = x.reshape(b, c, h, scale, w, scale)
    return v.permute(0, 1, 3, 5, 2, 4).reshape(b, c * scale * scale, h, w)


# ----------------------------------------------------------------------------- RRDBNet
def _conv3(sd, name, x):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=1, padding=1)


def _lrelu(x):
    return F.leaky_relu(x, LRELU_SLOPE)


def rdb(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    """ResidualDenseBlock.forward (SR/rrdbnet_arch.py:136-143)."""
    feats = [x]
    for k in range(1, 5):
        feats.append(_lrelu(_conv3(sd, f"{p}conv{k}", torch.cat(feats, 1))))
    x5 = _conv3(sd, f"{p}conv5", torch.cat(feats, 1))
    return x5 * RES_SCALE + x


def rrdb(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    """RRDB.forward (SR/rrdbnet_arch.py:162-167)."""
    out = x
    for k in (1, 2, 3):
        out = rdb(sd, f"{p}rdb{k}.", out)
    return out * RES_SCALE + x


def _rrdbnet_trunk(sd, x, scale, num_block):
    if scale == 2:
        feat = pixel_unshuffle(x, 2)
    elif scale == 1:
        feat = pixel_unshuffle(x, 4)
    else:
        feat = x
    feat = _conv3(sd, "conv_first", feat)
    body = feat
    for i in range(num_block):
        body = rrdb(sd, f"body.{i}.", body)
    feat = feat + _conv3(sd, "conv_body", body)
    feat = _lrelu(_conv3(sd, "conv_up1", nearest2x(feat)))
    feat = _lrelu(_conv3(sd, "conv_up2", nearest2x(feat)))
    return _conv3(sd, "conv_hr", feat)


def num_blocks_of(sd) -> int:
    n = 0
    while f"body.{n}.rdb1.conv1.weight" in sd:
        n += 1
    return n


@torch.no_grad()
def rrdbnet_forward_feature(sd, x, scale: int = 4) -> torch.Tensor:
    """RRDBNet.forward_feature (SR/rrdbnet_arch.py:225-240): NO activation after conv_hr."""
    return _rrdbnet_trunk(sd, x, scale, num_blocks_of(sd))


@torch.no_grad()
def rrdbnet_forward(sd, x, scale: int = 4) -> torch.Tensor:
    """RRDBNet.forward (SR/rrdbnet_arch.py:208-223)."""
    return _conv3(sd, "conv_last", _lrelu(_rrdbnet_trunk(sd, x, scale, num_blocks_of(sd))))


# ----------------------------------------------------------------------------- head (SR/HRfuse.py)
def upsampler(sd, p: str, x: torch.Tensor, scale: int = 4) -> torch.Tensor:
    """Upsampler (SR/HRfuse.py:17-44), power-of-two scales: [conv3x3 n->4n (bias), PixelShuffle(2)] x log2(scale);
    Sequential keys 0,2,4,... (odd indices are the PixelShuffle modules)."""
    if scale & (scale - 1) or scale < 2:
        raise NotImplementedError
    idx = 0
    s = scale
    while s > 1:
        x = pixel_shuffle(_conv3(sd, f"{p}{idx}", x), 2)
        idx += 2
        s //= 2
    return x


def _bn(sd, p, x, training):
    """nn.BatchNorm2d (train: batch statistics + running-stat update, eval: running stats)."""
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if training and (p + ".num_batches_tracked") in sd:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training, BN_MOMENTUM, BN_EPS)


def basic_block(sd, p: str, x: torch.Tensor, training: bool = False) -> torch.Tensor:
    """BasicBlock.forward (SR/HRfuse.py:142-159).  Running statistics in ``sd`` are updated in
    place when ``training`` (as nn.BatchNorm2d does)."""
    out = F.conv2d(x, sd[p + "conv1.weight"], None, 1, 1)
    out = F.relu(_bn(sd, p + "bn1", out, training))
    out = F.conv2d(out, sd[p + "conv2.weight"], None, 1, 1)
    out = _bn(sd, p + "bn2", out, training)
    if (p + "downsample.0.weight") in sd:
        idt = F.conv2d(x, sd[p + "downsample.0.weight"], None, 1, 0)
        idt = _bn(sd, p + "downsample.1", idt, training)
    else:
        idt = x
    return F.relu(out + idt)


def hrfeature(sd, p: str, x: torch.Tensor, training: bool = False) -> torch.Tensor:
    """HRfeature (SR/HRfuse.py:164-169): three BasicBlocks, Sequential keys 0,1,2."""
    for i in range(3):
        x = basic_block(sd, f"{p}{i}.", x, training)
    return x


def hrfuse_residual(sd, p: str, x_lr, x_hr, training: bool = False, upscale: int = 4):
    """HRfuse_residual.forward (SR/HRfuse.py:185-190)."""
    x_lr = upsampler(sd, p + "upsampler.", x_lr, upscale)
    x = torch.cat([x_lr, x_hr], 1)
    for i in range(3):
        x = basic_block(sd, f"{p}fuse.{i}.", x, training)
    return _conv3(sd, p + "conv_last", x)


# ----------------------------------------------------------------------------- aggregate
def aggregate_torch(data: torch.Tensor, scale: float) -> torch.Tensor:
    """aggregate_torch (aggregate_utils.py:29-41): step x step sum of data over the count of data>=0."""
    step = int(1 / scale)
    ones = torch.ones((1, 1, step, step), dtype=data.dtype)
    s1 = F.conv2d(data, ones, stride=step)
    s2 = F.conv2d((data >= 0).float(), ones, stride=step)
    return (s1 / (s2 + 1e-10)).squeeze()


# ----------------------------------------------------------------------------- error metrics used by every parity test
def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b|."""
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
