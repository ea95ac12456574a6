#!/bin/bash
# SR-stage: one-launch trunk weight gradients (srbh_trunk_wgrad): parity tests + A/B of the generator step
export TMPDIR=/tmp O=gpurun_out
timeout 1200 python -m pytest tests/test_sr_stage.py -m gpu -x -q 2>&1 | tail -8 | tee $O/r05bs_sr_tests.txt
(
for b in 8 24; do
  for v in 1 0; do
    echo "B=$b SRBH_SR_TRUNK_WGRAD=$v $(SRBH_SR_BENCH_MODES=fast SRBH_SR_TRUNK_WGRAD=$v timeout 600 python bench.py --workload sr_train --steps 10 --warmup 3 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config'].get('trunk_backward_calls'))")"
  done
done
) 2>&1 | tee $O/r05bs_sr_ab.txt
