# developer aid: tiled prediction (configs[4], first 12 cities) with fp32 vs fp16 activation tensors in the inference head, same box
run() { echo "$1: $(env $2 timeout 600 python bench.py --workload predict --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['p50_city_latency_ms'], d['large_cities'])")"; }
run "fp32 act" "SRBH_FP16_ACT=0"
run "fp16 act" "SRBH_FP16_ACT=1"
run "fp32 act" "SRBH_FP16_ACT=0"
run "fp16 act" "SRBH_FP16_ACT=1"
