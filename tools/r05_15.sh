#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
for v in 0 1; do echo "SRBH_HRFEAT_FIRST=$v"; SRBH_HRFEAT_FIRST=$v python tools/trunk_in_step_events.py 2>&1 | grep -v amdgpu; done | tee $O/r05p_trunk_in_step_events.txt
