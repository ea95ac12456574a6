#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
run() { echo "$* $(env "$@" timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; }
for r in 1 2 3; do
run SRBH_DEC2_SIDE=0
run SRBH_DEC2_SIDE=1
done 2>&1 | tee $O/r05bm_ab_dec2_side_again.txt
