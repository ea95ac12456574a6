# round 6: the fused eval BasicBlock after the counted-wait fixes: parity, time per block, predict A/B
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hblock16.py -q -x 2>&1 | tail -3 | tee $O/r06g_test_hblock16.txt
for wgs in 512 256 1024; do SRBH_HBLOCK16_WGS=$wgs timeout 200 python tools/time_hblock16.py 128 2>&1 | grep "fused"; done | tee $O/r06g_time_hblock16.txt
for v in 0 1 0 1; do SRBH_HBLOCK16=$v timeout 600 python bench.py --workload predict --steps 16 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('hblock16=$v', d['value'], d['p50_city_latency_ms'], d['head_paths_eager_calls'])"; done | tee $O/r06g_ab_predict_hblock16.txt
