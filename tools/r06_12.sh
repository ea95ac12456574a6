# round 6: how often does the strict-vs-mixed convergence A/B test fail (it failed once inside a full-suite run, r06j)?  the summary of 6 runs
O=gpurun_out; mkdir -p $O
python - <<'PY' 2>&1 | tee $O/r06l_convergence_ab_repeats.txt
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import convergence_ab as AB
dev = torch.device("cuda:0")
for rep in range(6):
    a = AB.run("f32", 120, 8, 1, 60, 1e-3, dev)
    b = AB.run("f16", 120, 8, 1, 60, 1e-3, dev)
    c = AB.run("f32", 120, 8, 1, 60, 1e-3, dev, drop_seed=4242)
    s = AB.summarise(a, b, tail=30, control=c)
    learn = [(r["mode"], round(sum(r["loss"][:10]) / 10, 2), round(sum(r["loss"][-30:]) / 30, 2), round(r["heldout_eval_height_rmse"][0][1], 3), round(r["heldout_eval_height_rmse"][-1][1], 3)) for r in (a, b, c)]
    print(json.dumps({"rep": rep, "mixed_over_strict": s["mixed_over_strict"], "noise": s["seed_noise_control"], "learn": learn}))
PY
