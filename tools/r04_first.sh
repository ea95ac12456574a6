#!/bin/bash
# round 4, first GPU pass: the new tests, the never-run configs[3] driver, the convergence A/B, the shrunk default line
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_epoch.py tests/test_gpu_convergence_ab.py tests/test_gpu_bench_ranks.py \
    "tests/test_sr_stage.py::test_fast_mode_two_forwards_before_backward_keep_their_own_saved_planes" \
    "tests/test_gpu_mbconv.py::test_bn_act_train_large_channel_offset" tests/test_gpu_mbconv.py::test_bn_act_train_matches_stock_ops 2>&1 | tail -15
timeout 600 python bench.py --workload epoch > $O/r04a_bench_epoch.json.log 2> $O/r04a_bench_epoch.err; echo "epoch rc=$?"; tail -c 1500 $O/r04a_bench_epoch.json.log
timeout 900 python tools/convergence_ab.py --steps 300 --batch 64 --out $O/r04a_convergence_ab.json > $O/r04a_convergence_ab.summary.json 2> $O/r04a_convergence_ab.err; echo "ab rc=$?"; cat $O/r04a_convergence_ab.summary.json
timeout 900 python bench.py --details $O/r04a_bench_details.json > $O/r04a_bench_default.json.log 2> $O/r04a_bench_default.err; echo "default rc=$?"; wc -c $O/r04a_bench_default.json.log; tail -c 2500 $O/r04a_bench_default.json.log
