#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
python tools/host_phase_time.py 10 prof 2>&1 | grep -v amdgpu.ids | tee $O/r05ai_host_phase_time.txt
