#!/bin/bash
# tiled prediction: trunk launches that leave CUs to the encoder / decoder graph on the second stream (SRBH_PT_IMAGES)
export TMPDIR=/tmp O=gpurun_out
for r in 1 2; do
for n in 0 30 28 24 20 16; do echo "SRBH_PT_IMAGES=$n $(SRBH_PT_IMAGES=$n timeout 600 python bench.py --workload predict --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d.get('p50_city_ms', d.get('latency',{})) )")"; done
done 2>&1 | tee $O/r05ad_ab_predict_pt_images.txt
