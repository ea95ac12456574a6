#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
t0=$(date +%s)
python bench.py > $O/r05bc_bench_default.json.log 2> $O/r05bc_bench_default.err
echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
tail -2 $O/r05bc_bench_default.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05bc_bench_default.json.log").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
t=d["train_step"]; print(t["ms_per_step"], t["value"], t["config"].get("pipeline","")[:90])
print(t.get("head_roofline",{}).get("frac_hbm_peak"), t.get("encdec_kernels",{}).get("ms_per_step"))
print(json.dumps(d["summary"])[:1100])
PY
