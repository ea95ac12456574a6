#!/bin/bash
# predict: wait for the encoder pass launched a batch ago BEFORE this batch's trunk (SRBH_PREDICT_LR_FIRST=1) instead of before reg / seg
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; TAG=${TAG:-r06al}
for r in 1 2 3; do
  for v in 0 1; do
    x=$(SRBH_PREDICT_LR_FIRST=$v timeout 900 python bench.py --workload predict --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('ms_per_step'))")
    echo "predict lr_first=$v $x" | tee -a $O/${TAG}_ab_lr_first.txt
  done
done
