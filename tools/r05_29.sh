#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_epoch.py tests/test_gpu_train_converges.py -x -q 2>&1 | tail -15
for r in 1 2; do
for v in 0 1; do echo "SRBH_TRAIN_PIPELINE=$v $(SRBH_TRAIN_PIPELINE=$v timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config'].get('pipeline','')[:60])")"; done
for n in 16 20 22 24; do echo "SRBH_PIPE_IMAGES=$n $(SRBH_PIPE_IMAGES=$n timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; done
done 2>&1 | tee $O/r05ac_ab_pipeline.txt
