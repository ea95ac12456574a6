"""torch.optim.Adam's update for the training step as ONE libsrbh launch (csrc/srbh_optim.hip).

The reference builds `torch.optim.Adam(net.parameters(), lr, weight_decay=1e-4)` and adds the three loss `log_var`s as a second parameter
group (train.py:170-179), stepping it once per batch (train.py:254-256).  `Adam` here is a torch.optim.Optimizer with the same constructor
arguments, parameter groups, `state` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter: `state_dict()` round-trips with torch's) and the
same arithmetic; only the execution differs: the moment buffers are views of two flat tensors, the tensor pointers sit in a device table
and `step()` is one kernel launch over a static list of 4096-element chunks instead of ~25 multi_tensor_apply launches (0.55 -> ~0.2 ms
per step of SRRegress_Cls_feature's 23 M parameters).  amsgrad / maximize / capturable / differentiable are not supported (the
reference uses none of them); CPU parameters fall back to torch.optim.Adam's own step.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib, wcache

_ENTRY = np.dtype([("p", np.uint64), ("g", np.uint64), ("m", np.uint64), ("v", np.uint64), ("n", np.int64), ("lr", np.float32), ("wd", np.float32),
                   ("inv_bc1", np.float32), ("inv_sqrt_bc2", np.float32)])


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("Adam: bad hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._plan = None          # (key, params, flat m, flat v, host table, device table, device chunk list, nchunks)
        self._t = 0

    # ---- the static part: which tensors, where their moments live, the chunk list
    def _build(self):
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if not ps:
            return None
        dev = ps[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 or not p.is_contiguous() for p in ps):
            raise NotImplementedError("srbh_amd.optim.Adam: contiguous fp32 parameters on one ROCm device")
        key = tuple((id(p), p.data_ptr(), p.numel()) for p in ps)
        if self._plan is not None and self._plan[0] == key:
            return self._plan
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4                 # (16-byte aligned slices: the kernel's vector path)
        m = torch.zeros(total, dtype=torch.float32, device=dev)
        v = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(ps, offs):                            # adopt existing state (load_state_dict, a re-plan after add_param_group)
            st = self.state[p]
            ma, va = m[o:o + p.numel()].view_as(p), v[o:o + p.numel()].view_as(p)
            if "exp_avg" in st:
                ma.copy_(st["exp_avg"])
                va.copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"] = ma, va
            st["step"] = float(st["step"]) if "step" in st else float(self._t)
        chunk = _lib.lib().srbh_adam_chunk()
        chunks = np.array([(i, s) for i, p in enumerate(ps) for s in range((p.numel() + chunk - 1) // chunk)], dtype=np.int32)
        # the pointer table goes to the device with an asynchronous copy from pinned memory; the host may be a whole step ahead of the
        # device, so the pinned image is a RING of three, each guarded by an event recorded behind its copy (reused only once that copy ran)
        ring = []
        for _ in range(3):
            host = torch.empty(len(ps) * _ENTRY.itemsize, dtype=torch.uint8).pin_memory()
            tab = host.numpy().view(_ENTRY)
            for i, (p, o) in enumerate(zip(ps, offs)):
                tab[i] = (p.data_ptr(), 0, m.data_ptr() + 4 * o, v.data_ptr() + 4 * o, p.numel(), 0.0, 0.0, 1.0, 1.0)
            ring.append([host, tab, None])
        dtabs = [torch.empty(ring[0][0].numel(), dtype=torch.uint8, device=dev) for _ in range(3)]
        dchunks = torch.from_numpy(chunks.reshape(-1)).to(dev)
        self._plan = (key, ps, m, v, ring, dtabs, dchunks, len(chunks))
        return self._plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plan = self._build()
        if plan is None:
            return loss
        _, ps, m, v, ring, dtabs, dchunks, nchunks = plan
        slot = self._t % 3
        host, tab, ev = ring[slot]
        dtab = dtabs[slot]
        if ev is not None:
            ev.synchronize()                               # (its copy of three steps ago: long done)
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        if any(g["betas"] != (b1, b2) or g["eps"] != eps for g in self.param_groups):
            raise NotImplementedError("srbh_amd.optim.Adam: one (betas, eps) for all parameter groups")
        i = 0
        any_grad = False
        bc_cache = {}
        for g in self.param_groups:
            lr, wd = float(g["lr"]), float(g["weight_decay"])
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                gr = p.grad
                if gr is not None and (gr.dtype != torch.float32 or not gr.is_contiguous() or gr.is_sparse):
                    raise NotImplementedError("srbh_amd.optim.Adam: dense contiguous fp32 gradients")
                tab["g"][i] = 0 if gr is None else gr.data_ptr()
                if gr is not None:
                    st = self.state[p]
                    t = int(st["step"]) + 1                # torch counts steps PER PARAMETER (one that got no gradient keeps its count)
                    st["step"] = float(t)                  # (a host number: torch.optim.Adam.__setstate__ turns it into its tensor on load)
                    bc = bc_cache.get(t)
                    if bc is None:
                        bc = bc_cache[t] = (1.0 / (1.0 - b1 ** t), 1.0 / math.sqrt(1.0 - b2 ** t))
                    tab["lr"][i], tab["wd"][i], tab["inv_bc1"][i], tab["inv_sqrt_bc2"][i] = lr, wd, bc[0], bc[1]
                    any_grad = True
                i += 1
        if not any_grad:
            return loss
        self._t += 1
        dtab.copy_(host, non_blocking=True)               # (pinned -> device on the current stream, in front of the launch)
        ev = torch.cuda.Event()
        ev.record()
        ring[slot][2] = ev
        L = _lib.lib()
        _lib.check(L.srbh_adam_step(dtab.data_ptr(), dchunks.data_ptr(), nchunks, b1, b2, eps, _lib.stream_ptr()), "adam_step")
        wcache.stamp(ps)                                   # weights changed behind the version counters (packed-weight caches, wcache.py)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plan = None                                  # the loaded moments are adopted into fresh flat buffers at the next step
        steps = [int(s["step"]) for s in self.state.values() if "step" in s]
        self._t = max(steps) if steps else 0

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._plan = None
