#!/bin/bash
# which LDS-tiled GEMM forms pay where (SRBH_PW_LDS_FORMS bit mask: 1 = 64x64, 2 = 64x128, 4 = 32x32 split K, 8 = 32x128)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; TAG=${TAG:-r06ac}
for r in 1 2; do
  for m in 0 15 11 4 3; do
    x=$(SRBH_PW_LDS_FORMS=$m timeout 900 python bench.py --workload predict --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('ms_per_step'))")
    echo "predict forms=$m $x" >> $O/${TAG}_ab_pw_lds_forms.txt
    x=$(SRBH_PW_LDS_FORMS=$m timeout 900 python bench.py --workload train --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('ms_per_step'))")
    echo "train forms=$m $x" >> $O/${TAG}_ab_pw_lds_forms.txt
  done
done
sort -s -k1,2 $O/${TAG}_ab_pw_lds_forms.txt
