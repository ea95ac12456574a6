# round 6: hbwd16_kernel with counted waits (unconditional loads, constant prefetch issue, skip-gradient loads always issued): 0 B of scratch in all
# six forms (were 8-72 B).  Tests, then same-box A/B against the previous kernel (build/variants/libsrbh_hbwd16orig.so)
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hbwd16.py tests/test_gpu_head_f16.py tests/test_gpu_io16.py tests/test_gpu_head.py -q -x 2>&1 | tail -4 | tee $O/r06o_tests_hbwd16.txt
for r in 1 2; do for v in orig new; do L=""; [ $v = orig ] && L=build/variants/libsrbh_hbwd16orig.so; SRBH_LIB_PATH=$L timeout 300 python tools/time_hbwd16.py 2>&1 | grep -v "^$" | sed "s/^/$v /" | tail -8; done; done | tee $O/r06o_time_hbwd16.txt
run() { SRBH_TRAIN_PIPELINE=$3 SRBH_LIB_PATH=$2 timeout 300 python bench.py --workload train --steps 30 --warmup 8 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('train pipe=$3 $1', d['ms_per_step'])"; }
for r in 1 2 3; do run orig build/variants/libsrbh_hbwd16orig.so 0; run new "" 0; run orig build/variants/libsrbh_hbwd16orig.so 1; run new "" 1; done | tee $O/r06o_ab_hbwd16_waits.txt
