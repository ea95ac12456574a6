#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
run() { echo "$* $(env "$@" timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; }
for r in 1 2; do
for u in 0 1 2 4 8; do run SRBH_HOST_SPIN_US=$u; done
done 2>&1 | tee $O/r05ah_host_spin_pipelined.txt
