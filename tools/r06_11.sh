# round 6: the two tests that failed in r06j on their own (full output), predict at larger batches
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_feature_h16.py tests/test_gpu_convergence_ab.py -q 2>&1 | tail -40 | tee $O/r06k_two_tests.txt
for b in 128 256 384; do timeout 600 python bench.py --workload predict --steps 24 --warmup 2 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('batch $b', d['value'], d['p50_city_latency_ms'], d['p95_city_latency_ms'])"; done | tee $O/r06k_predict_batch.txt
