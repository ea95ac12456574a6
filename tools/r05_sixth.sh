#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hbwd16.py -x -q 2>&1 | tail -30
timeout 1500 python -m pytest tests/test_gpu_head.py tests/test_gpu_head_f16.py tests/test_gpu_io16.py tests/test_gpu_model.py tests/test_gpu_train_converges.py tests/test_gpu_graph_lifetime.py -x -q 2>&1 | tail -8
bash tools/ab_env.sh train SRBH_HBWD16=0 SRBH_HBWD16=1 2>&1 | tee $O/r05f_ab_hbwd16.txt
