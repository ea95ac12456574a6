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
    _steps_dirty = False
    _aux = None
    _plan = None

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("Adam: bad hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._plan = None          # (key, params, flat m, flat v, host table, device table, device chunk list, nchunks)
        self._aux = None           # [plan, per-parameter step counts, (lr, wd, n) per group, lr per tensor, wd per tensor]
        self._steps_dirty = False
        self._t = 0

    # ---- the static part: which tensors, where their moments live, the chunk list
    def _build(self):
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if not ps:
            return None
        key = tuple((id(p), p.data_ptr(), p.numel()) for p in ps)
        if self._plan is not None and self._plan[0] == key:
            return self._plan
        dev = ps[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 or not p.is_contiguous() for p in ps):
            raise NotImplementedError("srbh_amd.optim.Adam: contiguous fp32 parameters on one ROCm device")
        self._sync_steps()                                     # (a re-plan adopts state[p]["step"]: bring it up to date first)
        self._aux = None
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
            # a parameter without state starts at step 0 as in torch/optim/adam.py (one added by add_param_group, or unfrozen, in the
            # middle of training has its OWN bias corrections; `_t` only picks the ring slot)
            st["step"] = float(st["step"]) if "step" in st else 0.0
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
        # The per-step part of the table in array form (this loop is host time in front of the step's last launch: written element by
        # element through the structured array it took 1.8 ms for SRRegress_Cls_feature's ~700 tensors -- tools/step_boundary_probe.py)
        aux = self._aux
        if aux is None or aux[0] is not plan:
            steps = np.array([float(self.state[p].get("step", 0.0)) for p in ps], dtype=np.float64)
            aux = self._aux = [plan, steps, None, None, None]
        steps = aux[1]
        hp = tuple((float(g["lr"]), float(g["weight_decay"]), sum(1 for p in g["params"] if p.requires_grad)) for g in self.param_groups)
        if aux[2] != hp:                                   # (lr / weight decay per tensor: rebuilt when a scheduler changes a group)
            aux[2] = hp
            aux[3] = np.concatenate([np.full(n, lr, dtype=np.float32) for lr, _, n in hp])
            aux[4] = np.concatenate([np.full(n, wd, dtype=np.float32) for _, wd, n in hp])
        ptrs = np.zeros(len(ps), dtype=np.uint64)
        f32 = torch.float32
        for i, p in enumerate(ps):
            gr = p.grad
            if gr is not None:
                if gr.dtype is not f32 or gr.is_sparse or not gr.is_contiguous():
                    raise NotImplementedError("srbh_amd.optim.Adam: dense contiguous fp32 gradients")
                ptrs[i] = gr.data_ptr()
        has = ptrs != 0
        if not has.any():
            return loss
        steps[has] += 1.0                                  # torch counts steps PER PARAMETER (one that got no gradient keeps its count)
        t = np.maximum(steps, 1.0)
        tab["g"] = ptrs
        tab["lr"] = aux[3]
        tab["wd"] = aux[4]
        tab["inv_bc1"] = (1.0 / (1.0 - np.power(b1, t))).astype(np.float32)
        tab["inv_sqrt_bc2"] = (1.0 / np.sqrt(1.0 - np.power(b2, t))).astype(np.float32)
        self._steps_dirty = True                           # (state[p]["step"] is brought up to date when somebody looks: _sync_steps)
        self._t += 1
        dtab.copy_(host, non_blocking=True)               # (pinned -> device on the current stream, in front of the launch)
        ev = torch.cuda.Event()
        ev.record()
        ring[slot][2] = ev
        L = _lib.lib()
        _lib.check(L.srbh_adam_step(dtab.data_ptr(), dchunks.data_ptr(), nchunks, b1, b2, eps, _lib.stream_ptr()), "adam_step")
        wcache.stamp(ps)                                   # weights changed behind the version counters (packed-weight caches, wcache.py)
        return loss

    def _sync_steps(self):
        """per-parameter step counts live in one array between steps; `state[p]["step"]` (torch's layout) is refreshed when somebody looks
        (the `state` property below: `optimizer.state[p]`, `state_dict()`, a re-plan)"""
        aux = getattr(self, "_aux", None)
        dirty, self._steps_dirty = getattr(self, "_steps_dirty", False), False
        if dirty and aux is not None and self._plan is not None and aux[0] is self._plan:
            st = self._state
            for p, t in zip(self._plan[1], aux[1]):
                st[p]["step"] = float(t)                   # (a host number: torch.optim.Adam.__setstate__ turns it into its tensor on load)

    @property
    def state(self):
        if self._steps_dirty:
            self._sync_steps()
        return self._state

    @state.setter
    def state(self, value):
        self._state = value

    def __setstate__(self, state):
        super().__setstate__(state)                        # (torch updates __dict__ directly: move a loaded 'state' behind the property)
        if "state" in self.__dict__:
            self._state = self.__dict__.pop("state")

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._aux, self._steps_dirty = None, False
        self._plan = None                                  # the loaded moments are adopted into fresh flat buffers at the next step
        steps = [int(s["step"]) for s in self.state.values() if "step" in s]
        self._t = max(steps) if steps else 0

    def add_param_group(self, param_group):
        if getattr(self, "_plan", None) is not None:
            self._sync_steps()
        super().add_param_group(param_group)
        self._aux = None
        self._plan = None
