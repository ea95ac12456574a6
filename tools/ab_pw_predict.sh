# developer aid: tiled prediction with the libsrbh 1x1 convolutions on (SRBH_PWCONV=1) / off in the encoder (default: training only)
run() { SRBH_PWCONV=$1 timeout 400 python bench.py --workload predict --steps 12 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pwconv=$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do run train; run 1; done
