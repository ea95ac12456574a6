#!/bin/bash
# per-kernel durations of the three parts of a prediction batch, each alone on the device (B = 256)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out
for p in lr fuse hr; do
  rm -rf /tmp/pk_$p
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk_$p -- python tools/predict_part_kernels.py $p 256 6 > /dev/null 2>&1
  echo "== $p (7 runs incl. the first; divide totals by 7)" >> $O/r06u_predict_part_kernels.txt
  python tools/pw_trace.py /tmp/pk_$p "" >> $O/r06u_predict_part_kernels.txt
done
