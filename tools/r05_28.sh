#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
python tools/partition_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r05ab_partition_probe.txt
