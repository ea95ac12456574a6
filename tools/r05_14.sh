#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
bash tools/ab_env.sh train SRBH_HOST_SPIN_US=0 SRBH_HOST_SPIN_US=3 SRBH_HOST_SPIN_US=6 SRBH_HOST_SPIN_US=12 2>&1 | tee $O/r05o_host_spin.txt
