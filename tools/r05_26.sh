#!/bin/bash
# hbwd16 BS = 2: the ReLU bit words by one vector load ahead of the MFMAs instead of 16 scalar loads in the epilogue
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_hbwd16.py -x -q 2>&1 | tail -4
for r in 1 2; do
echo "== vector bits"; python tools/time_hbwd16.py 2>&1 | grep -v amdgpu.ids | tail -6
echo "== scalar bits"; SRBH_LIB_PATH=build/variants/libsrbh_bitsv0.so python tools/time_hbwd16.py 2>&1 | grep -v amdgpu.ids | tail -6
done 2>&1 | tee $O/r05z_time_hbwd16_bits.txt
