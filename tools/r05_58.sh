#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
(
echo "== HEAD"; timeout 600 python tools/soak_pipelined.py 600 2
echo "== HEAD, SRBH_PT_WT=1"; SRBH_PT_WT=1 timeout 600 python tools/soak_pipelined.py 600 2
echo "== P3_PREREAD=0"; SRBH_LIB_PATH=build/variants/libsrbh_pr0.so timeout 600 python tools/soak_pipelined.py 600 2
echo "== P3_WFLAGS=0 P3_PREREAD=0"; SRBH_LIB_PATH=build/variants/libsrbh_wf0.so timeout 600 python tools/soak_pipelined.py 600 2
echo "== P3_SEAM=0"; SRBH_LIB_PATH=build/variants/libsrbh_seam0.so timeout 600 python tools/soak_pipelined.py 600 2
) 2>&1 | grep -v amdgpu.ids | tee $O/r05bo_soak_pipelined_variants.txt
