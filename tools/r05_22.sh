#!/bin/bash
# the hand-over across a layer boundary (P3_PRELAYER): parity tests, same-box A/B (base = PRELAYER 2, PRE_AT 2)
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_rrdbnet.py tests/test_gpu_feature_h16.py -x -q 2>&1 | tail -6
bash tools/ab_variants.sh pl0 pl1 preat3 preat1 2>&1 | tee $O/r05v_ab_prelayer.txt
