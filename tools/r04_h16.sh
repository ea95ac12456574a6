#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_feature_h16.py tests/test_gpu_head_f16.py tests/test_gpu_model.py tests/test_gpu_rrdbnet.py tests/test_gpu_graph_lifetime.py 2>&1 | tail -15
for v in 1 0; do
  SRBH_FEATURE_H16=$v timeout 600 python bench.py --workload train --steps 10 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('train FEATURE_H16=$v', d['ms_per_step'], d['value'])"
  SRBH_FEATURE_H16=$v timeout 600 python bench.py --workload predict --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('predict FEATURE_H16=$v', d['value'], d['p50_city_latency_ms'])"
done
SRBH_HCONV_ENTRY_WGS=512 timeout 600 python bench.py --workload train --steps 10 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('train entry wgs=512', d['ms_per_step'], d['value'])"
