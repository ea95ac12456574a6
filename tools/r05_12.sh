#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
bash tools/ab_env.sh train SRBH_HRFEAT_FIRST=0 SRBH_HRFEAT_FIRST=1 2>&1 | tee $O/r05m_ab_hrfeat_first.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train_converges.py -x -q 2>&1 | tail -4
