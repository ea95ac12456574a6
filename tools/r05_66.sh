#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 600 python -m pytest tests/test_sr_stage.py -m gpu -x -q -k "one_launch or persistent_training_backward" 2>&1 | tail -3
(for b in 8 24; do python tools/time_trunk_wgrad.py $b; done) 2>&1 | grep -v amdgpu.ids | tee $O/r05bw_trunk_wgrad.txt
