# round 6: the trunk's chunk-3/4 steps as one do-while body + polls only where a check follows: full GPU suite, A/B against the rolled-with-dead-break build
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/r06b_gpu_tests.txt; cat $O/r06b_gpu_tests.txt
bash tools/ab_variants.sh rolled 2>&1 | tee $O/r06b_ab_c34_dowhile.txt
