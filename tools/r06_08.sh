# round 6: hconv16_kernel with counted waits (unconditional loads, constant prefetch issue, pre-loop consumption): head tests, then same-box
# A/B against the previous kernel (build/variants/libsrbh_hc16orig.so): train step and tiled prediction
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_head.py tests/test_gpu_head_f16.py tests/test_gpu_io16.py tests/test_gpu_hbwd16.py tests/test_gpu_hblock16.py tests/test_gpu_model.py -q -x 2>&1 | tail -4 | tee $O/r06h_tests_head.txt
run() { SRBH_LIB_PATH=$2 timeout 300 python bench.py --workload train --steps 20 --warmup 8 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('train $1', d['ms_per_step'])"; }
runp() { SRBH_LIB_PATH=$2 timeout 300 python bench.py --workload predict --steps 16 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('predict $1', d['value'])"; }
for r in 1 2; do run orig build/variants/libsrbh_hc16orig.so; run new ""; done | tee $O/r06h_ab_hconv16_waits.txt
for r in 1 2; do runp orig build/variants/libsrbh_hc16orig.so; runp new ""; done | tee -a $O/r06h_ab_hconv16_waits.txt
