#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
{
python tools/power_ablate.py base
python tools/power_ablate.py base zeros
for t in abl1 abl2 abl4 abl3 abl7; do SRBH_LIB_PATH=build/variants/libsrbh_$t.so python tools/power_ablate.py $t zeros; done
} 2>&1 | grep -v amdgpu.ids | tee $O/r04c_trunk_power_ablation.txt
