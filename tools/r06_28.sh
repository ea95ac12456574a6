#!/bin/bash
# ptail_kernel: where a tile's time goes -- ablation builds (timing only, wrong results): 1 = no MFMAs, 2 = no input DMA after the first tile, 4 = no stores
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; TAG=${TAG:-r06ag}
run() { SRBH_LIB_PATH=$2 timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], 'outside the trunk:', round(d['ms_per_step']-d['roofline']['avg_launch_ms'],4))"; }
for r in 1 2; do run base ""; for a in 1 2 4; do run abl$a build/variants/libsrbh_ptabl$a.so; done; done | tee $O/${TAG}_ptail_ablation.txt
