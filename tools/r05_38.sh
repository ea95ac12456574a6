#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_hbwd16.py tests/test_gpu_head_f16.py tests/test_gpu_head.py -x -q 2>&1 | tail -5
run() { echo "$* $(env "$@" timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; }
for r in 1 2 3; do
run SRBH_WGRAD_DEFER=0
run SRBH_WGRAD_DEFER=1
done 2>&1 | tee $O/r05ak_ab_wgrad_defer.txt
