"""developer aid / evidence for BENCH's `train_step.parity`: the parameter gradients of ONE full training step in the mixed mode
(head_precision="f16": fp16 forward operands, bf16 gradient operands, fp32 accumulation -- harness.TrainStep's default) against the same step in
the exact-fp32 mode, same weights, same batch, lr = 0, drop-connect off.  Per parameter group and for the whole gradient vector: rel-L2, cosine,
and the share of the squared gradient norm carried by tensors within the stated tolerance.  usage: grad_parity.py [batch] [num_block] [seed]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def grads(headp, batch, num_block, seed, dev):
    from srbh_amd import encoders, synth
    from srbh_amd.harness import TrainStep, synthetic_batch
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    encoders.DROP_CONNECT = 0.0
    net_hr = RRDBNet(3, 3, num_block=num_block)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=num_block, seed=1337, mode="init"))
    torch.manual_seed(seed)
    net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
    ts = TrainStep(net_hr.to(dev), net.to(dev), dev, lr=0.0, status_every=0, head_precision=headp)
    b = synthetic_batch(batch, 4242, dev)
    loss, _ = ts(b)
    torch.cuda.synchronize()
    names = [k for k, _ in net.named_parameters()] + ["log_var0", "log_var1", "log_var2"]
    return float(loss), {k: (None if p.grad is None else p.grad.detach().double().cpu()) for k, p in zip(names, ts.params())}


def compare(ga, gb, tol):
    groups, tot = {}, [0.0, 0.0, 0.0, 0.0]      # per group: sum |a-b|^2, sum |b|^2, sum a.b, sum |a|^2
    within = 0.0
    for k, b in gb.items():
        a = ga[k]
        if b is None:
            assert a is None, k
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
               "share_of_grad_norm2": round(g[1] / tot[1], 6)} for k, g in groups.items()}
    whole = {"rel_l2": round((tot[0] / tot[1]) ** 0.5, 6), "cos": round(tot[2] / (tot[1] * tot[3]) ** 0.5, 6),
             "norm_share_of_tensors_within_tol": round(within / tot[1], 6), "tol": tol}
    return out, whole


if __name__ == "__main__":
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 23
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1337
    dev = torch.device("cuda", 0)
    l16, g16 = grads("f16", batch, nb, seed, dev)
    l32, g32 = grads("f32", batch, nb, seed, dev)
    l32b, g32b = grads("f32", batch, nb, seed, dev)      # the exact mode against itself: the run-to-run floor (BatchNorm atomics)
    for tol in (5e-3, 1e-2, 2e-2, 5e-2):
        groups, whole = compare(g16, g32, tol)
        print(json.dumps({"batch": batch, "loss_f16": l16, "loss_f32": l32, "whole": whole}))
    print(json.dumps({"groups_f16_vs_f32": groups}))
    groups0, whole0 = compare(g32b, g32, 5e-3)
    print(json.dumps({"floor_f32_vs_f32": whole0, "groups": groups0}))
