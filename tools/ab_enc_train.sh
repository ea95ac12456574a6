# developer aid: same-box A/B of the training step with the libsrbh training BatchNorm / squeeze-excite kernels on (default) and off
run() { SRBH_ENC_TRAIN_FUSED=$1 timeout 400 python bench.py --workload train --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fused=$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do run 1; run 0; done
