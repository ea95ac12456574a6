#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
(
for b in 8 24; do
echo "== B=$b"; python tools/time_trunk_wgrad.py $b
for v in 1 2 4 6 7; do echo "TW_ABL=$v: $(SRBH_LIB_PATH=build/variants/libsrbh_twabl$v.so python tools/time_trunk_wgrad.py $b)"; done
done
) 2>&1 | grep -v amdgpu.ids | tee $O/r05bv_trunk_wgrad_ablation.txt
