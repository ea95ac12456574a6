# round 6: hconv_entry_kernel with counted waits (0 B of scratch, was 100 B): tests, same-box A/B against the previous kernel
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_head_f16.py tests/test_gpu_feature_h16.py tests/test_gpu_head.py tests/test_gpu_model.py tests/test_gpu_hblock16.py -q -x 2>&1 | tail -4 | tee $O/r06p_tests_entry.txt
run() { SRBH_TRAIN_PIPELINE=$3 SRBH_LIB_PATH=$2 timeout 300 python bench.py --workload train --steps 30 --warmup 8 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('train pipe=$3 $1', d['ms_per_step'])"; }
runp() { SRBH_LIB_PATH=$2 timeout 300 python bench.py --workload predict --steps 24 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('predict $1', d['value'])"; }
for r in 1 2 3; do run orig build/variants/libsrbh_entryorig.so 0; run new "" 0; run orig build/variants/libsrbh_entryorig.so 1; run new "" 1; runp orig build/variants/libsrbh_entryorig.so; runp new ""; done | tee $O/r06p_ab_entry_waits.txt
