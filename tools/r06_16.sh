# round 6: bn_add_relu with four units per trip (all loads in front of the arithmetic): tests, kernel time by the bench's own per-call table, A/B
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_head.py tests/test_gpu_head_f16.py tests/test_gpu_io16.py tests/test_gpu_hbwd16.py -q -x 2>&1 | tail -3 | tee $O/r06q_tests_bar.txt
run() { SRBH_TRAIN_PIPELINE=$3 SRBH_LIB_PATH=$2 timeout 300 python bench.py --workload train --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline())
bar=[r for r in d['head_roofline']['kernels'] if 'bn_add_relu' in r['kernel']]
print('train pipe=$3 $1', d['ms_per_step'], 'head', d['head_roofline']['ms_per_step'], d['head_roofline']['frac_hbm_peak'], 'bn_add_relu', bar[0]['us_per_call'] if bar else None)"; }
for r in 1 2 3; do run orig build/variants/libsrbh_barorig.so 1; run new "" 1; done | tee $O/r06q_ab_bar_unroll.txt
