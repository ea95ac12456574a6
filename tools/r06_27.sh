#!/bin/bash
# dw_lds_eval_kernel: up to 8 rounds of planes per workgroup at large batch (SRBH_DW_EVAL_ROUNDS): tests, encoder graph time, predict A/B
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; TAG=${TAG:-r06ae}
timeout 1500 python -m pytest tests/test_gpu_dwconv.py tests/test_gpu_model.py tests/test_gpu_graph_lifetime.py -x -q -m gpu > $O/${TAG}_tests_dw_rounds.txt 2>&1; tail -3 $O/${TAG}_tests_dw_rounds.txt
for r in 1 2; do
  for v in 1 2 4 8; do
    echo "rounds=$v: $(SRBH_DW_EVAL_ROUNDS=$v python tools/predict_parts.py 256 20 2>/dev/null | grep '^batch' | tail -1 | cut -c1-60) | $(SRBH_DW_EVAL_ROUNDS=$v python tools/predict_parts.py 256 20 2>/dev/null | grep '^whole' | tail -1 | cut -c40-110)" >> $O/${TAG}_predict_parts.txt
  done
done
cat $O/${TAG}_predict_parts.txt
for r in 1 2 3; do
  for v in 1 8; do
    x=$(SRBH_DW_EVAL_ROUNDS=$v timeout 900 python bench.py --workload predict --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('ms_per_step'))")
    echo "predict rounds=$v $x" >> $O/${TAG}_ab_dw_rounds.txt
  done
done
cat $O/${TAG}_ab_dw_rounds.txt
