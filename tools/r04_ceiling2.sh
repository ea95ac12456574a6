#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
python tools/dump_trunk_operands.py $O/trunk_acts.bin $O/trunk_weights.bin
( while true; do echo "t=$(date +%s.%N)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk'; sleep 0.5; done ) > $O/r04k_mfma_ceiling_smi.log 2>&1 &
SMI=$!
CEIL_SECONDS=5 timeout 600 tools/mfma_ceiling $O/trunk_acts.bin $O/trunk_weights.bin 2>&1 | while IFS= read -r line; do echo "t=$(date +%s.%N) $line"; done | tee $O/r04k_mfma_ceiling.txt
kill $SMI
rm -f $O/trunk_acts.bin $O/trunk_weights.bin
