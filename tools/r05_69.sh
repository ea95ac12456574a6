#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 1500 python -m pytest tests/test_sr_stage.py -m gpu -x -q 2>&1 | tail -3
(echo "== discriminator convs on libsrbh"; timeout 600 python tools/sr_iteration_phases.py 8; echo "== SRBH_SR_DISC=stock"; SRBH_SR_DISC=stock timeout 600 python tools/sr_iteration_phases.py 8) 2>&1 | grep -v amdgpu.ids | tee $O/r05cg_sr_iteration_phases.txt
