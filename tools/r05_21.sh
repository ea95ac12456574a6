#!/bin/bash
# the step barrier inside the step's last group + the next step's first reads behind it (P3_PREREAD): parity tests, same-box A/B
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_rrdbnet.py tests/test_gpu_feature_h16.py -x -q 2>&1 | tail -6
bash tools/ab_variants.sh pre0 preat2 2>&1 | tee $O/r05u_ab_preread.txt
