#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
cd /tmp
for b in 8 24; do
  rm -rf /tmp/srt; SRBH_SR_BENCH_ITERATION=0 SRBH_SR_BENCH_MODES=fast timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/srt -- python /root/repo/bench.py --workload sr_train --steps 6 --warmup 2 --batch $b > /dev/null 2>&1
  f=$(find /tmp/srt -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-200 > /root/repo/$O/r05cl_sr_train_b${b}_kernel_stats.csv; head -7 /root/repo/$O/r05cl_sr_train_b${b}_kernel_stats.csv | cut -c1-150
done
