#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 300 python -m pytest -x -q -m gpu tests/test_gpu_dconv.py 2>&1 | tail -3
timeout 300 python tools/time_dconv.py 128 find 2>&1 | grep -v amdgpu.ids | tee $O/r04g_time_dconv_b128_find.txt
for v in 1 0; do
  SRBH_DCONV=$v timeout 600 python bench.py --workload predict --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('predict DCONV=$v', d['value'], d['p50_city_latency_ms'])"
  SRBH_SIDE_STREAM=0 SRBH_DCONV=$v timeout 600 python bench.py --workload predict --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('predict one-stream DCONV=$v', d['value'], d['p50_city_latency_ms'])"
done
