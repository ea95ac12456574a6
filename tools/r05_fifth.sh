#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sidework.py -x -q 2>&1 | tail -3
bash tools/ab_train_lib.sh pwuk16 pwuk32 2>&1 | tee $O/r05e_ab_pw_uk.txt
