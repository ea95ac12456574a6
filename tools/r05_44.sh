#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
run() { echo "$* $(env "$@" timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; }
for r in 1 2 3 4; do
run SRBH_PIPE_TAIL_WGS=256
run SRBH_PIPE_TAIL_WGS=192
run SRBH_PIPE_TAIL_WGS=160
done 2>&1 | tee $O/r05ap_ab_pipe_tail_wgs.txt
