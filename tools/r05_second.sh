#!/bin/bash
# round 5, second GPU call: side-stream weight gradients -- tests, then the same-box A/B of the training step
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sidework.py tests/test_gpu_mbconv.py tests/test_gpu_dconv.py tests/test_gpu_dp_trainstep.py -x -q 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_bench_ranks.py -x -q -k exact_driver 2>&1 | tail -8
bash tools/ab_env.sh train SRBH_WGRAD_SIDE=0 SRBH_WGRAD_SIDE=1 2>&1 | tee $O/r05b_ab_wgrad_side.txt
