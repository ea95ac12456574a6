#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sidework.py -x -q 2>&1 | tail -4
bash tools/ab_env.sh train SRBH_TRAIN_GRAPH=0 SRBH_TRAIN_GRAPH=1 2>&1 | tee $O/r05c_ab_train_graph.txt
SRBH_WGRAD_SIDE=0 bash tools/ab_env.sh train SRBH_TRAIN_GRAPH=0 SRBH_TRAIN_GRAPH=1 2>&1 | tee -a $O/r05c_ab_train_graph.txt
