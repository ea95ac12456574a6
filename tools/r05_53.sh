#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
SRBH_SR_BENCH_MODES=fast timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/srt -- python bench.py --workload sr_train --steps 6 --warmup 2 > /dev/null 2>&1
f=$(find /tmp/srt -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-200 | tee $O/r05bg_sr_train_b8_fast_kernel_stats.csv
