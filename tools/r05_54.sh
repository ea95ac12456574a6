#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 1500 python -m pytest tests/test_sr_stage.py -x -q 2>&1 | tail -3
for r in 1 2 3; do
for v in 0 1; do echo "SRBH_SR_BATCH_REDUCE=$v $(SRBH_SR_BENCH_MODES=fast SRBH_SR_BATCH_REDUCE=$v timeout 600 python bench.py --workload sr_train --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; done
done 2>&1 | tee $O/r05bh_ab_sr_batch_reduce.txt
for v in 0 1; do echo "B=24 SRBH_SR_BATCH_REDUCE=$v $(SRBH_SR_BENCH_MODES=fast SRBH_SR_BATCH_REDUCE=$v timeout 600 python bench.py --workload sr_train --steps 6 --warmup 2 --batch 24 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; done 2>&1 | tee -a $O/r05bh_ab_sr_batch_reduce.txt
