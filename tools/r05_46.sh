#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_mbconv.py tests/test_gpu_pipeline.py tests/test_gpu_train_converges.py -x -q 2>&1 | tail -4
run() { echo "$* $(env "$@" timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; }
for r in 1 2 3 4; do
run SRBH_PW_WGRAD_DEFER=0
run SRBH_PW_WGRAD_DEFER=1
run SRBH_PW_WGRAD_DEFER=0 SRBH_TRAIN_PIPELINE=0
run SRBH_PW_WGRAD_DEFER=1 SRBH_TRAIN_PIPELINE=0
done 2>&1 | tee $O/r05as_ab_pw_wgrad_defer.txt
