#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
for v in 0 1; do SRBH_TRAIN_PIPELINE=$v timeout 900 python bench.py --workload train --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pipeline $v', d['ms_per_step'], 'strict', d['strict_f32_head']['ms_per_step'])"; done 2>&1 | tee $O/r05bl_strict_pipelined.txt
