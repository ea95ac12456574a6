"""Strict-vs-mixed convergence A/B of the BASELINE configs[2] training step (round-3 VERDICT, What's weak #1).

Two runs of harness.TrainStep from the SAME initial weights over the SAME sequence of device-drawn batches (labels are a function of
the tile: harness.learnable_batch_device; same drop-connect draws), one with head_precision='f32' (exact-fp32 head convolutions and
gradients: the mode the <=5e-5 gradient-parity tests pin), one with 'f16' (fp16 forward / bf16 gradient operands: TrainStep's default).
Per step: training loss and height RMSE of the training batch; every `--eval-every` steps the eval-mode height RMSE on 4 held-out
batches.  Writes one JSON (curves + summary).

    python tools/convergence_ab.py --steps 300 --batch 64 --out gpurun_out/convergence_ab.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(mode, steps, batch, num_block, eval_every, lr, dev, drop_seed=None):
    from srbh_amd import synth
    from srbh_amd.harness import TrainStep, learnable_batch_device
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    net_hr = RRDBNet(3, 3, num_block=num_block)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=num_block, seed=1337, mode="init"))
    torch.manual_seed(1337)                                   # same initial weights AND the same drop-connect draws in both runs
    net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
    if drop_seed is not None:                                # control run: same weights, same batches, OTHER drop-connect draws
        torch.manual_seed(drop_seed)
    ts = TrainStep(net_hr.to(dev), net.to(dev), dev, lr=lr, head_precision=mode, status_every=0)
    gen = torch.Generator(device=dev)
    held = []
    for i in range(4):
        gen.manual_seed(999_000 + i)
        held.append(learnable_batch_device(batch, gen, dev))

    def evaluate():
        from srbh_amd import hrfuse as H
        net.eval()
        se, n = 0.0, 0
        with torch.no_grad(), H.head_precision(mode):
            for b in held:
                hp = net(b[0], net_hr.forward_feature(b[0][:, :3]))[0].squeeze(1)
                se += float(((hp - b[1]) ** 2).sum())
                n += b[1].numel()
        net.train()
        return (se / n) ** 0.5

    loss_curve, rmse_curve, evals = [], [], [[0, evaluate()]]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        gen.manual_seed(1337 + 104729 * i)
        b = learnable_batch_device(batch, gen, dev)
        loss, hp = ts(b)
        loss_curve.append(float(loss))
        rmse_curve.append(float(((hp.squeeze(1) - b[1]) ** 2).mean().sqrt()))
        if (i + 1) % eval_every == 0:
            evals.append([i + 1, evaluate()])
    torch.cuda.synchronize()
    return {"mode": mode, "loss": loss_curve, "train_height_rmse": rmse_curve, "heldout_eval_height_rmse": evals,
            "wall_s": round(time.perf_counter() - t0, 2)}


def summarise(a, b, tail=50, control=None):
    """a = strict, b = mixed, control = strict again with other drop-connect draws (the seed-noise yardstick: two runs of the SAME
    arithmetic differ by this much after `steps` steps)"""
    mean = lambda v: sum(v) / len(v)                      # noqa: E731
    if control is not None:
        s = summarise(a, b, tail)
        c = summarise(a, control, tail)
        s["seed_noise_control"] = {"what": "strict f32 vs strict f32 with other drop-connect draws", "loss_tail_mean": c["loss_tail_mean"],
                                   "loss_tail_rel_gap": c["loss_tail_rel_gap"], "train_rmse_tail_rel_gap": c["train_rmse_tail_rel_gap"],
                                   "heldout_rmse_final": c["heldout_rmse_final"], "heldout_rmse_final_rel_gap": c["heldout_rmse_final_rel_gap"]}
        s["mixed_over_strict"] = {"loss_tail": round(s["loss_tail_mean"][1] / s["loss_tail_mean"][0], 4),
                                  "train_rmse_tail": round(s["train_rmse_tail_mean"][1] / s["train_rmse_tail_mean"][0], 4),
                                  "heldout_rmse_final": round(s["heldout_rmse_final"][1] / s["heldout_rmse_final"][0], 4)}
        return s
    la, lb = mean(a["loss"][-tail:]), mean(b["loss"][-tail:])
    ra, rb = mean(a["train_height_rmse"][-tail:]), mean(b["train_height_rmse"][-tail:])
    ea, eb = a["heldout_eval_height_rmse"][-1][1], b["heldout_eval_height_rmse"][-1][1]
    return {"tail_steps": tail,
            "loss_first10": [round(mean(a["loss"][:10]), 5), round(mean(b["loss"][:10]), 5)],
            "loss_tail_mean": [round(la, 5), round(lb, 5)], "loss_tail_rel_gap": round(abs(la - lb) / abs(la), 5),
            "train_rmse_tail_mean": [round(ra, 4), round(rb, 4)], "train_rmse_tail_rel_gap": round(abs(ra - rb) / ra, 5),
            "heldout_rmse_start": [round(a["heldout_eval_height_rmse"][0][1], 4), round(b["heldout_eval_height_rmse"][0][1], 4)],
            "heldout_rmse_final": [round(ea, 4), round(eb, 4)], "heldout_rmse_final_rel_gap": round(abs(ea - eb) / ea, 5),
            "order": ["f32 (strict)", "f16 (mixed)"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--num-block", type=int, default=23)
    ap.add_argument("--eval-every", type=int, default=50)
    ap.add_argument("--lr", type=float, default=1e-3)        # train.py:153
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    a = run("f32", args.steps, args.batch, args.num_block, args.eval_every, args.lr, dev)
    torch.cuda.empty_cache()
    b = run("f16", args.steps, args.batch, args.num_block, args.eval_every, args.lr, dev)
    torch.cuda.empty_cache()
    c = run("f32", args.steps, args.batch, args.num_block, args.eval_every, args.lr, dev, drop_seed=4242)
    out = {"config": vars(args), "summary": summarise(a, b, tail=min(50, args.steps // 2), control=c), "strict_f32": a, "mixed_f16": b,
           "strict_f32_other_dropconnect_seed": c}
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f)
    print(json.dumps({"config": out["config"], "summary": out["summary"], "wall_s": [a["wall_s"], b["wall_s"], c["wall_s"]]}))


if __name__ == "__main__":
    main()
