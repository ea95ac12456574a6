#!/bin/bash
# LDS-tiled 1x1-conv GEMM on / off (SRBH_PW_LDS): training step and tiled prediction, interleaved
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; TAG=${TAG:-r06ab}
for r in 1 2 3; do
  for v in 0 1; do
    x=$(SRBH_PW_LDS=$v timeout 900 python bench.py --workload train --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('ms_per_step'), d.get('serial_step',{}).get('ms_per_step'))")
    echo "train pw_lds=$v $x" >> $O/${TAG}_ab_pw_lds.txt
    x=$(SRBH_PW_LDS=$v timeout 900 python bench.py --workload predict --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('ms_per_step'))")
    echo "predict pw_lds=$v $x" >> $O/${TAG}_ab_pw_lds.txt
  done
done
cat $O/${TAG}_ab_pw_lds.txt
