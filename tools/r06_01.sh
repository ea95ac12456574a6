# round 6, first GPU pass: the pruned build -- full GPU suite, same-box A/B of the trunk's chunk-3/4 steps (two bodies = in-tree vs one rolled body),
# the default bench line (configs[4] at all 301 cities inside it)
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/r06a_gpu_tests.txt; cat $O/r06a_gpu_tests.txt
bash tools/ab_variants.sh rolled 2>&1 | tee $O/r06a_ab_c34_bodies.txt
timeout 1200 python bench.py > $O/r06a_bench_default.json.log 2> $O/r06a_bench_default.err; tail -c 3000 $O/r06a_bench_default.json.log
