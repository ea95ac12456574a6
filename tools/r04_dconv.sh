#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_dconv.py tests/test_gpu_feature_h16.py 2>&1 | tail -15
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_model.py tests/test_gpu_mbconv.py tests/test_gpu_train_converges.py tests/test_gpu_predict_sharded.py 2>&1 | tail -5
timeout 300 python tools/time_dconv.py 64 2>&1 | grep -v amdgpu.ids | tee $O/r04g_time_dconv_b64.txt
for v in 1 0; do
  SRBH_DCONV=$v timeout 600 python bench.py --workload train --steps 10 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('train DCONV=$v', d['ms_per_step'], d['value'])"
  SRBH_DCONV=$v timeout 600 python bench.py --workload predict --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('predict DCONV=$v', d['value'], d['p50_city_latency_ms'])"
done
