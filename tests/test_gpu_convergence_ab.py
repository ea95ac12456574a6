"""GPU: strict-vs-mixed convergence A/B (round-3 VERDICT, What's weak #1).  TrainStep's default head precision ('f16': fp16 forward /
bf16 gradient operands) has gradients that are only direction-accurate against the exact graph (tests/test_gpu_head_f16.py); what a
user needs to know is whether it TRAINS like the exact-fp32 head.  Same weights, same varying batches with learnable labels, same
drop-connect draws, both modes -- plus a strict-mode control run with OTHER drop-connect draws as the noise yardstick: the loss has
to fall in all three, and the mixed mode must not end worse than the strict mode beyond that noise.  The 300-step, batch-64,
23-block run of the same tool is kept in profiles/ (r04*_convergence_ab.json)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_mixed_head_trains_like_strict_head():
    import convergence_ab as AB
    dev = torch.device("cuda:0")
    steps = 120
    a = AB.run("f32", steps, 8, 1, 60, 1e-3, dev)
    b = AB.run("f16", steps, 8, 1, 60, 1e-3, dev)
    c = AB.run("f32", steps, 8, 1, 60, 1e-3, dev, drop_seed=4242)       # seed-noise control: the SAME arithmetic, other drop-connect draws
    s = AB.summarise(a, b, tail=30, control=c)
    for r in (a, b, c):
        assert all(v == v and v < 1e9 for v in r["loss"])
        first, last = sum(r["loss"][:10]) / 10, sum(r["loss"][-30:]) / 30
        assert last < 0.5 * first, (r["mode"], first, last)                  # it learns (labels depend on the tile)
        assert r["heldout_eval_height_rmse"][-1][1] < 0.8 * r["heldout_eval_height_rmse"][0][1]
    # Training is chaotic: two runs of the exact-fp32 step that differ only in their drop-connect draws end 10-20 % apart (the
    # control), so "same curve" can only mean "inside that band".  One-sided: the mixed mode must not train WORSE than the strict
    # mode by more than the larger of 50 % and twice the control's gap (it may train better: first run here, 300 steps at B=64:
    # loss tail 5.94 vs 6.88, held-out RMSE 4.91 vs 5.80 -- noise in its favour).
    # (round 6, profiles/r06l_convergence_ab_repeats.txt: six repeats of exactly this test body -- the strict-vs-strict control gap of the loss tail
    #  ranged 0.05 ... 0.44, the mixed / strict ratio 0.83 ... 1.19: ONE control sample underestimates the band one run in ~ten, which is how
    #  this test failed once inside a full-suite run.  The floor is therefore the largest control gap seen, 0.5, not 0.25.)
    noise = s["seed_noise_control"]
    m = s["mixed_over_strict"]
    assert m["loss_tail"] <= 1.0 + max(0.5, 2 * noise["loss_tail_rel_gap"]), s
    assert m["train_rmse_tail"] <= 1.0 + max(0.5, 2 * noise["train_rmse_tail_rel_gap"]), s
    assert m["heldout_rmse_final"] <= 1.0 + max(0.5, 2 * noise["heldout_rmse_final_rel_gap"]), s
