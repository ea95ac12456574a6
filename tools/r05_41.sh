#!/bin/bash
# RRDB-closing epilogue: three rows of the RRDB-level stream by LDS-DMA at the start of the epilogue (P3_R2LDS): parity tests, same-box A/B, timeline
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_rrdbnet.py tests/test_gpu_feature_h16.py -x -q 2>&1 | tail -6
bash tools/ab_variants.sh r2lds0 2>&1 | tee $O/r05an_ab_r2lds.txt
bash tools/prof_variants.sh r2lds0 2>&1 | grep -v amdgpu | tee -a $O/r05an_ab_r2lds.txt
