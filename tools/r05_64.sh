#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 600 python -m pytest tests/test_sr_stage.py -m gpu -x -q -k "one_launch or persistent_training_backward" 2>&1 | tail -3
cd /tmp
for b in 8 24; do
  rm -rf /tmp/srt; SRBH_SR_BENCH_MODES=fast timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/srt -- python /root/repo/bench.py --workload sr_train --steps 6 --warmup 2 --batch $b > /tmp/b.json 2>/dev/null
  f=$(find /tmp/srt -name "*kernel_stats.csv" | head -1); grep "trunk_wgrad\|ptrunk3" $f | cut -c1-150
  python -c "import json; d=json.loads(open('/tmp/b.json').readline()); print('B=$b', d['ms_per_step'], d['roofline']['achieved'])"
done 2>&1 | tee /root/repo/$O/r05bu_wgrad_xcd.txt
