#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hbwd16.py -x -q 2>&1 | tail -30
timeout 1500 python -m pytest tests/test_gpu_head.py tests/test_gpu_head_f16.py tests/test_gpu_io16.py tests/test_gpu_model.py tests/test_gpu_train_converges.py tests/test_gpu_graph_lifetime.py tests/test_gpu_syncbn.py tests/test_gpu_dp_trainstep.py -x -q 2>&1 | tail -8
bash tools/ab_env.sh train SRBH_CHAIN_HANDOFF=0 SRBH_CHAIN_HANDOFF=1 SRBH_BLOCK_CHAIN=0 2>&1 | tee $O/r05j_ab_chain.txt
