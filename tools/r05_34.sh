#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
python tools/step_boundary_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r05ag_step_boundary.txt
