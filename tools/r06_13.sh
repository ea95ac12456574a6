# round 6: predict batch size, interleaved repeats on one box (12 + 24 cities)
O=gpurun_out; mkdir -p $O
for r in 1 2 3; do for b in 128 256 192; do timeout 600 python bench.py --workload predict --steps 24 --warmup 2 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('batch $b', d['value'], d['p50_city_latency_ms'], d['p95_city_latency_ms'], d['tail_shapes_run'])"; done; done | tee $O/r06m_predict_batch_repeats.txt
