#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
python tools/launch_cost_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r05aa_launch_cost_probe.txt
