#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_head.py tests/test_gpu_head_f16.py tests/test_gpu_hbwd16.py tests/test_gpu_io16.py tests/test_gpu_syncbn.py -x -q 2>&1 | tail -3
for r in 1 2 3; do
echo "old finalize: $(SRBH_LIB_PATH=build/variants/libsrbh_finold.so python tools/time_head_defer.py 2>&1 | grep -v amdgpu | tail -1)"
echo "new finalize: $(SRBH_LIB_PATH=build/variants/libsrbh_finnew.so python tools/time_head_defer.py 2>&1 | grep -v amdgpu | tail -1)"
done 2>&1 | tee $O/r05au_time_head_finalize.txt
