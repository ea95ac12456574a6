#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest -q -m gpu tests/test_gpu_convergence_ab.py tests/test_gpu_bench_ranks.py \
    "tests/test_sr_stage.py::test_fast_mode_two_forwards_before_backward_keep_their_own_saved_planes" \
    "tests/test_gpu_mbconv.py::test_bn_act_train_large_channel_offset" tests/test_gpu_mbconv.py::test_bn_act_train_matches_stock_ops 2>&1 | tail -25
timeout 900 python tools/convergence_ab.py --steps 300 --batch 64 --out $O/r04b_convergence_ab.json > $O/r04b_convergence_ab.summary.json 2> $O/r04b_convergence_ab.err; echo "ab rc=$?"; cat $O/r04b_convergence_ab.summary.json
# MFMA ceiling under the power cap, with rocm-smi sampled beside it
python tools/dump_trunk_operands.py $O/trunk_acts.bin $O/trunk_weights.bin
( while true; do echo "t=$(date +%s.%N)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk'; sleep 0.5; done ) > $O/r04b_mfma_ceiling_smi.log 2>&1 &
SMI=$!
CEIL_SECONDS=6 timeout 600 tools/mfma_ceiling $O/trunk_acts.bin $O/trunk_weights.bin 2>&1 | while IFS= read -r line; do echo "t=$(date +%s.%N) $line"; done | tee $O/r04b_mfma_ceiling.txt
kill $SMI
rm -f $O/trunk_acts.bin $O/trunk_weights.bin
