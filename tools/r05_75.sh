#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_sr_stage.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
(for b in 8 24; do echo "B=$b $(SRBH_SR_BENCH_ITERATION=0 SRBH_SR_BENCH_MODES=fast timeout 600 python bench.py --workload sr_train --steps 10 --warmup 3 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")"; done) | tee $O/r05cq_sr_lrelu_bwd.txt
