# developer aid: tiled prediction (configs[4], first 12 cities) under inference-head storage variants, same box
run() { echo "$1: $(env $2 timeout 600 python bench.py --workload predict --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['p50_city_latency_ms'], d['large_cities'])")"; }
run "fp16 act, hrfeat out fp16" "SRBH_FP16_ACT=1 SRBH_HRFEAT_OUT_H16=1"
run "fp16 act, hrfeat out fp32" "SRBH_FP16_ACT=1 SRBH_HRFEAT_OUT_H16=0"
run "fp16 act, hrfeat out fp16" "SRBH_FP16_ACT=1 SRBH_HRFEAT_OUT_H16=1"
run "fp16 act, hrfeat out fp32" "SRBH_FP16_ACT=1 SRBH_HRFEAT_OUT_H16=0"
