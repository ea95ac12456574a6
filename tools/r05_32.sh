#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
run() { echo "$* $(env "$@" timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; }
for r in 1 2; do
run SRBH_TRAIN_PIPELINE=0
run SRBH_PIPE_IMAGES=16
run SRBH_PIPE_IMAGES=24
run SRBH_PIPE_IMAGES=16 SRBH_PIPE_AT=reg
run SRBH_PIPE_IMAGES=24 SRBH_PIPE_AT=reg
run SRBH_PIPE_IMAGES=32 SRBH_PIPE_AT=reg
done 2>&1 | tee $O/r05ae_ab_pipeline2.txt
