#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
run() { echo "$* $(env "$@" timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; }
for r in 1 2; do
run SRBH_PIPE_IMAGES=16
run SRBH_PIPE_IMAGES=24
run SRBH_PIPE_IMAGES=32
run SRBH_PIPE_IMAGES=16 SRBH_PIPE_TAIL_WGS=256
run SRBH_PIPE_IMAGES=16 SRBH_PIPE_TAIL_WGS=128
run SRBH_PIPE_IMAGES=16 SRBH_PIPE_AT=reg
run SRBH_PIPE_IMAGES=8
done 2>&1 | tee $O/r05bf_pipeline_resweep.txt
