"""aggregate_torch on the GPU (reference aggregate_utils.py:29-41): step x step block sum of ``data`` over the
count of ``data >= 0`` (+1e-10), then ``.squeeze()``.  One libsrbh kernel; no CPU fallback."""
from __future__ import annotations

import torch

from . import _lib

__all__ = ["aggregate_torch"]


def aggregate_torch(data, scale):
    if not (torch.is_tensor(data) and data.is_cuda):
        raise RuntimeError("aggregate_torch (libsrbh): input must be a ROCm/HIP device tensor (no CPU fallback; the "
                           "DataLoader-side CPU use of the reference stays with the caller)")
    if data.dim() != 4 or data.shape[1] != 1:
        raise ValueError(f"expected a (N,1,H,W) tensor as in the reference, got {tuple(data.shape)}")
    step = int(1 / scale)
    n, _, h, w = data.shape
    x = data.detach().float().contiguous()
    out = torch.empty((n, 1, h // step, w // step), dtype=torch.float32, device=data.device)
    _lib.check(_lib.lib().srbh_aggregate(x.data_ptr(), out.data_ptr(), n, h, w, step, _lib.stream_ptr()), "aggregate")
    return out.squeeze()
