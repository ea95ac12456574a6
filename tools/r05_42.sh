#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
run() { echo "$* $(env "$@" timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; }
for r in 1 2 3; do
run SRBH_PTAIL_WGS=0
run SRBH_PTAIL_WGS=128
run SRBH_PTAIL_WGS=192
done 2>&1 | tee $O/r05ao_ab_ptail_wgs.txt
