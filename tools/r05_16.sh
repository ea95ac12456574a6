#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_adam.py -x -q 2>&1 | tail -12
timeout 1200 python -m pytest tests/test_gpu_train_converges.py tests/test_gpu_dp_trainstep.py tests/test_gpu_fullsize.py tests/test_gpu_epoch.py tests/test_gpu_convergence_ab.py -x -q 2>&1 | tail -6
bash tools/ab_env.sh train SRBH_ADAM=0 SRBH_ADAM=1 2>&1 | tee $O/r05q_ab_adam.txt
