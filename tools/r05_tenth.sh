#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
echo "LATE=1 (in-tree)"; python tools/time_hbwd16.py 64 2>&1 | grep -v amdgpu | tee $O/r05k_time_hbwd16_chain.txt
for l in 0 2; do echo "LATE=$l"; SRBH_LIB_PATH=build/variants/libsrbh_hblate$l.so python tools/time_hbwd16.py 64 2>&1 | grep -v amdgpu | head -1 | tee -a $O/r05k_time_hbwd16_chain.txt; done
