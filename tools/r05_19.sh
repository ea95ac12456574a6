#!/bin/bash
# round 5, re-entry: where the training step stands at HEAD (steady kernel table + gaps), the headline line
export TMPDIR=/tmp O=gpurun_out TAG=r05s
mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/${TAG}_bench_feature_b32.json.log 2>/tmp/e1 || tail -5 /tmp/e1
timeout 600 python bench.py --workload train --no-cpu-baseline > $O/${TAG}_bench_train_b64.json.log 2>/tmp/e2 || tail -5 /tmp/e2
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/${TAG}_tr -- python bench.py --workload train --steps 6 --warmup 3 --no-extras > /dev/null 2>&1
python tools/steady_stats.py /tmp/${TAG}_tr 4 80 --stock > $O/${TAG}_train_steady_kernel_stats.txt
python tools/gap_stats.py /tmp/${TAG}_tr 4 12 >> $O/${TAG}_train_steady_kernel_stats.txt
tail -1 $O/${TAG}_bench_feature_b32.json.log | cut -c1-600
tail -1 $O/${TAG}_bench_train_b64.json.log | cut -c1-900
