#!/bin/bash
# per-wave neighbour checks inside the fetching step (P3_WFLAGS): parity tests, then the same-box A/B against -DP3_WFLAGS=0
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_rrdbnet.py tests/test_gpu_feature_h16.py -x -q 2>&1 | tail -6
bash tools/ab_variants.sh wflags0 2>&1 | tee $O/r05t_ab_wflags.txt
