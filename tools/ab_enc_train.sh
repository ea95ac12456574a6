# developer aid: same-box A/B of the training step: libsrbh training BatchNorm / squeeze-excite kernels and 1x1 convolutions on (default) / off
run() { SRBH_ENC_TRAIN_FUSED=$1 SRBH_PWCONV=$2 timeout 400 python bench.py --workload train --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fused=$1 pwconv=$2', d['value'], d['ms_per_step'])"; }
for r in 1 2; do run 1 train; run 1 0; run 0 0; done
