#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
for w in 512 384 256 128; do echo "WGS=$w"; SRBH_HBWD16_WGS=$w python tools/time_hbwd16.py 64 2>&1 | grep -v amdgpu; done | tee $O/r05g_time_hbwd16.txt
