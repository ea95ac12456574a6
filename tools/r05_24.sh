#!/bin/bash
# centre-tap fragments from the register-resident planes (P3_CTAP): parity tests, same-box A/B
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_rrdbnet.py tests/test_gpu_feature_h16.py -x -q 2>&1 | tail -6
bash tools/ab_variants.sh ctap0 2>&1 | tee $O/r05x_ab_ctap.txt
