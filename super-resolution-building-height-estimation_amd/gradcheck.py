"""Parameter-gradient parity of the training step's precision modes (bench.py `train_step.parity`, tests/test_gpu_grad_parity.py,
tools/grad_parity.py).

The reference trains in fp32 through torch autograd (train.py:243-257).  harness.TrainStep's default mode ("f16": fp16 forward operands, bf16
data / weight gradient operands, fp32 accumulation, fp32 BatchNorm / losses / Adam) is compared here with the exact-fp32 mode on the SAME
weights and batch: one step each with lr = 0 and drop-connect off, then per parameter group (first component of the parameter name) and for
the whole gradient vector: rel-L2, cosine and the share of the squared gradient norm carried by tensors within a tolerance."""
from __future__ import annotations

import torch

# stated tolerances of the mixed mode against the exact graph (measured at batch 64 / 23 blocks and batch 8 / 2 blocks, profiles/r06c_grad_parity_*):
# the heads' own parameters (`reg`, `seg`, `aggre_height`, the loss `log_var`s) see exact-fp32 loss gradients through at most three bf16-operand
# convolutions; everything upstream of them (`hrfeat`, `decoder1/2`, `encoder`) receives its gradient through the heads' ~20 bf16-rounded
# data-gradient stages and ~100 training-mode BatchNorms: direction-accurate, not digit-accurate
TOL = {"whole_rel_l2": 2.5e-2, "heads_rel_l2": 4e-3, "upstream_rel_l2": 1.5e-1, "upstream_cos": 0.99}
HEAD_GROUPS = ("reg", "seg", "aggre_height", "log_var0", "log_var1", "log_var2")


def step_gradients(net_hr, net, batch, dev, head_precision):
    """{parameter name: float64 CPU gradient or None} of ONE TrainStep (lr 0) in `head_precision`; the nets are left unchanged"""
    from . import encoders
    from .harness import TrainStep
    old = encoders.DROP_CONNECT
    encoders.DROP_CONNECT = 0.0
    try:
        ts = TrainStep(net_hr, net, dev, lr=0.0, status_every=0, head_precision=head_precision)
        loss, _ = ts(batch)
        torch.cuda.synchronize()
    finally:
        encoders.DROP_CONNECT = old
    names = [k for k, _ in net.named_parameters()] + ["log_var0", "log_var1", "log_var2"]
    grads = {k: (None if p.grad is None else p.grad.detach().double().cpu()) for k, p in zip(names, ts.params())}
    for p in ts.params():
        p.grad = None
    return float(loss), grads


def compare(ga, gb, tol=5e-3):
    """(per group, whole): ga against the yardstick gb"""
    groups, tot, within = {}, [0.0, 0.0, 0.0, 0.0], 0.0
    for k, b in gb.items():
        a = ga[k]
        if b is None:
            if a is not None:
                raise AssertionError(f"{k}: a gradient where the exact graph has none")
            continue
        d2, b2, ab, a2 = float((a - b).pow(2).sum()), float(b.pow(2).sum()), float((a * b).sum()), float(a.pow(2).sum())
        g = groups.setdefault(k.split(".")[0], [0.0, 0.0, 0.0, 0.0, 0])
        for i, v in enumerate((d2, b2, ab, a2)):
            g[i] += v
            tot[i] += v
        g[4] += 1
        if b2 > 0 and (d2 / b2) ** 0.5 <= tol:
            within += b2
    out = {k: {"tensors": g[4], "rel_l2": round((g[0] / max(g[1], 1e-300)) ** 0.5, 6), "cos": round(g[2] / max((g[1] * g[3]) ** 0.5, 1e-300), 6),
               "share_of_grad_norm2": round(g[1] / max(tot[1], 1e-300), 6)} for k, g in groups.items()}
    whole = {"rel_l2": round((tot[0] / max(tot[1], 1e-300)) ** 0.5, 6), "cos": round(tot[2] / max((tot[1] * tot[3]) ** 0.5, 1e-300), 6),
             "norm2_share_of_tensors_within": round(within / max(tot[1], 1e-300), 6), "within_tol": tol}
    return out, whole


def mixed_vs_exact(net_hr, net, batch, dev):
    """the parity object of bench.py: mixed ("f16") against exact ("f32") gradients of the same step, + the exact mode against itself
    (run-to-run floor: BatchNorm partial sums are added with atomics)"""
    l16, g16 = step_gradients(net_hr, net, batch, dev, "f16")
    l32, g32 = step_gradients(net_hr, net, batch, dev, "f32")
    _, g32b = step_gradients(net_hr, net, batch, dev, "f32")
    groups, whole = compare(g16, g32)
    _, floor = compare(g32b, g32)
    heads = max(v["rel_l2"] for k, v in groups.items() if k in HEAD_GROUPS)
    up = {k: v for k, v in groups.items() if k not in HEAD_GROUPS}
    ok = (whole["rel_l2"] <= TOL["whole_rel_l2"] and heads <= TOL["heads_rel_l2"]
          and all(v["rel_l2"] <= TOL["upstream_rel_l2"] and v["cos"] >= TOL["upstream_cos"] for v in up.values()))
    return {"what": "parameter gradients of one training step, head_precision 'f16' (fp16 forward / bf16 gradient operands, fp32 accumulation) against "
                    "'f32' (exact fp32 matrix cores), same weights and batch, lr 0, drop-connect off",
            "loss": {"f16": l16, "f32": l32, "rel": round(abs(l16 - l32) / max(abs(l32), 1e-30), 9)},
            "whole_gradient": whole, "exact_mode_run_to_run": {"rel_l2": floor["rel_l2"]},
            "heads_max_rel_l2": heads, "groups": {k: {kk: v[kk] for kk in ("rel_l2", "cos", "share_of_grad_norm2")} for k, v in groups.items()},
            "tolerance": TOL, "within_tolerance": bool(ok)}
