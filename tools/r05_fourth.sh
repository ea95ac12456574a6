#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sidework.py -x -q 2>&1 | tail -30
timeout 1500 python -m pytest tests/test_gpu_head.py tests/test_gpu_head_f16.py tests/test_gpu_io16.py tests/test_gpu_syncbn.py tests/test_gpu_dp_trainstep.py tests/test_gpu_graph_lifetime.py tests/test_gpu_model.py tests/test_gpu_feature_h16.py tests/test_gpu_train_converges.py -x -q 2>&1 | tail -12
bash tools/ab_env.sh train SRBH_STATS_POOL=0 SRBH_STATS_POOL=1 2>&1 | tee $O/r05d_ab_stats_pool.txt
