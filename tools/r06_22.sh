#!/bin/bash
# predict: next batch's encoder / decoders behind this batch's trunk (SRBH_PREDICT_AHEAD) -- tests, graph timing, bench A/B
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; TAG=${TAG:-r06y}
timeout 1500 python -m pytest tests/test_gpu_graph_lifetime.py tests/test_gpu_model.py -x -q -m gpu > $O/${TAG}_tests_predict_ahead.txt 2>&1; tail -3 $O/${TAG}_tests_predict_ahead.txt
timeout 600 python tools/predict_parts.py 256 20 2>&1 | tail -5 > $O/${TAG}_predict_parts.txt; cat $O/${TAG}_predict_parts.txt
for r in 1 2 3; do
  for a in 0 1; do
    v=$(SRBH_PREDICT_AHEAD=$a timeout 900 python bench.py --workload predict --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('ms_per_step'))")
    echo "ahead=$a $v" >> $O/${TAG}_ab_predict_ahead.txt
  done
done
cat $O/${TAG}_ab_predict_ahead.txt
