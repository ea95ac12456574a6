#!/bin/bash
# tiled prediction: CUDA-graph replay (default) vs the eager loop after this round's host diet, interleaved, 30-city sample
export TMPDIR=/tmp O=gpurun_out
run() { timeout 600 python bench.py --workload predict --steps 30 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['p50_city_latency_ms'], d['p95_city_latency_ms'], d['max_city_latency_ms'])"; }
(
echo "# tiles/s, p50 / p95 / max city latency (ms)"
for r in 1 2; do
echo "graph: $(run)"
echo "eager: $(SRBH_PREDICT_GRAPH=0 run)"
done
) 2>&1 | tee $O/r05ck_ab_predict_graph_vs_eager.txt
