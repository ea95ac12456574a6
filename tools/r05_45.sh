#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
run() { echo "$* $(env "$@" timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")"; }
for r in 1 2 3 4 5 6 7 8; do
run SRBH_PIPE_TAIL_WGS=256 SRBH_PIPE_PRIO=0
run SRBH_PIPE_TAIL_WGS=256 SRBH_PIPE_PRIO=-1
run SRBH_PIPE_TAIL_WGS=256 GPU_MAX_HW_QUEUES=8
done 2>&1 | tee $O/r05aq_pipeline_modes.txt
