#!/usr/bin/env python3
"""Per-depth error budget of the fp16-operand fast path (round-4 VERDICT item 3).

`RRDBNet.forward_feature` (reference SR/rrdbnet_arch.py:225-240) on nets TRUNCATED to their first k RRDB blocks (same seeded
weights: the per-key generators of srbh_amd.synth make body.{0..k-1} of the k-block net equal to those of the 23-block net),
fast path vs the strict-fp32 GPU path (`net.precision = "f32"`: exact-fp32 matrix cores, pinned to ~1e-6 of the CPU oracle by
tests/test_gpu_rrdbnet.py::test_strict_fp32_path...).  Prints the cumulative rel-L2 of the 64 x 256 x 256 feature map after
k = 1, 2, 4, ... 23 blocks (tail convs included in every row: the k -> k+1 increment is the trunk's contribution), for several
weight seeds and both weight modes, and writes a JSON summary.

    python tools/layer_error_budget.py [out.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    from srbh_amd import synth
    from srbh_amd.rrdbnet import RRDBNet
    dev = "cuda:0"
    depths = (1, 2, 4, 8, 12, 16, 20, 23)
    seeds = (1337, 1, 2, 3)
    x = synth.tiles(1, 8, 64, seed=1337)[:, :3].contiguous().to(dev)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())      # noqa: E731
    out = {"depths": list(depths), "rows": []}
    for mode in ("init", "stress"):
        for ws in seeds:
            row = []
            for k in depths:
                sd = synth.rrdbnet_state_dict(num_block=k, seed=ws, mode=mode)
                net = RRDBNet(3, 3, num_block=k)
                net.load_state_dict(sd, strict=True)
                net = net.to(dev).eval()
                with torch.no_grad():
                    y_fast = net.forward_feature(x).float()
                    net.precision = "f32"
                    y_strict = net.forward_feature(x).float()
                net.check_status()
                row.append(rel(y_fast, y_strict))
                del net
            out["rows"].append({"mode": mode, "weight_seed": ws, "rel_l2_after_k_blocks": row})
            print(f"{mode:6s} seed {ws:5d}: " + " ".join(f"k={k}:{e:.2e}" for k, e in zip(depths, row)), flush=True)
    worst = max(max(r["rel_l2_after_k_blocks"]) for r in out["rows"])
    out["max_rel_l2"] = worst
    print(f"max over all rows: {worst:.3e} (tolerance 1e-3)")
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
