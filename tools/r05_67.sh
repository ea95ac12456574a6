#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
(for s in 1 2 3 4 6 8; do for b in 8 24; do echo "splits=$s $(SRBH_TWG_SPLITS=$s python tools/time_trunk_wgrad.py $b)"; done; done) 2>&1 | grep -v amdgpu.ids | tee $O/r05by_trunk_wgrad_splits.txt
