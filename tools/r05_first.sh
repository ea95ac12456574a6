#!/bin/bash
# round 5, first GPU call: the new tests, the parity budget, the operand-order sweep of the MFMA ceiling, the default bench line
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_sweep.py tests/test_gpu_feature_h16.py tests/test_gpu_head_f16.py -x -q -s 2>&1 | tail -25 > $O/r05a_tests_new.txt
timeout 1700 python -m pytest tests/test_gpu_bench_ranks.py -x -q 2>&1 | tail -15 >> $O/r05a_tests_new.txt
timeout 900 python tools/layer_error_budget.py $O/r05a_layer_error_budget.json > $O/r05a_layer_error_budget.txt 2>&1
python tools/dump_trunk_operands.py $O/trunk_acts.bin $O/trunk_weights.bin
CEIL_SECONDS=4 CEIL_SKIP_DUTY=1 timeout 900 tools/mfma_ceiling $O/trunk_acts.bin $O/trunk_weights.bin > $O/r05a_mfma_ceiling_orders.txt 2>&1
rm -f $O/trunk_acts.bin $O/trunk_weights.bin
timeout 900 python bench.py --details $O/r05a_bench_details.json > $O/r05a_bench_feature_b32.json.log 2> $O/r05a_bench.err
tail -5 $O/r05a_bench.err
cat $O/r05a_tests_new.txt
tail -3 $O/r05a_layer_error_budget.txt
cat $O/r05a_mfma_ceiling_orders.txt
tail -c 1500 $O/r05a_bench_feature_b32.json.log
