#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 1500 python -m pytest tests/test_sr_stage.py tests/test_gpu_rrdbnet.py tests/test_gpu_feature_h16.py -x -q 2>&1 | tail -5
for r in 1 2; do
echo "feature $(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")"
for v in 0 1; do echo "SRBH_SR_PTRUNK=$v $(SRBH_SR_BENCH_MODES=fast SRBH_SR_PTRUNK=$v timeout 600 python bench.py --workload sr_train --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; done
done 2>&1 | tee $O/r05bi_ab_sr_ptrunk.txt
for v in 0 1; do echo "B=24 SRBH_SR_PTRUNK=$v $(SRBH_SR_BENCH_MODES=fast SRBH_SR_PTRUNK=$v timeout 600 python bench.py --workload sr_train --steps 6 --warmup 2 --batch 24 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; done 2>&1 | tee -a $O/r05bi_ab_sr_ptrunk.txt
