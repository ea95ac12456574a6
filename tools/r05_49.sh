#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_head_f16.py tests/test_gpu_head.py tests/test_gpu_train_converges.py tests/test_gpu_pipeline.py tests/test_gpu_graph_lifetime.py tests/test_gpu_dp_trainstep.py -x -q 2>&1 | tail -4
run() { echo "$* $(env "$@" timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; }
for r in 1 2 3; do
run SRBH_PACK_AFTER_STEP=0
run SRBH_PACK_AFTER_STEP=1
run SRBH_PACK_AFTER_STEP=0 SRBH_TRAIN_PIPELINE=0
run SRBH_PACK_AFTER_STEP=1 SRBH_TRAIN_PIPELINE=0
done 2>&1 | tee $O/r05av_ab_pack_after_step.txt
