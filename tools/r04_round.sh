#!/bin/bash
export TMPDIR=/tmp
bash tools/profile_round.sh r04m 2>&1 | tail -40
bash tools/ab_train_lib.sh entry2 2>&1 | tail -6
python tools/host_phase_time.py 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r04m_host_phase_time.txt; tail -12 gpurun_out/r04m_host_phase_time.txt
