# round 6: ptail_kernel leaves the previous tile's epilogue stores in flight at the top-of-tile wait (vmcnt(16 | 32 | 48) instead of vmcnt(0)):
# conv / feature tests, same-box A/B (feature bench: whole forward ms; predict)
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_rrdbnet.py tests/test_gpu_feature_h16.py -q -x 2>&1 | tail -3 | tee $O/r06t_tests_ptail.txt
run() { SRBH_LIB_PATH=$2 timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['ms_per_step']-d['roofline']['avg_launch_ms'],4))"; }
for r in 1 2 3 4; do run orig build/variants/libsrbh_ptailorig.so; run new ""; done | tee $O/r06t_ab_ptail_waits.txt
runp() { SRBH_LIB_PATH=$2 timeout 300 python bench.py --workload predict --steps 24 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('predict $1', d['value'])"; }
for r in 1 2; do runp orig build/variants/libsrbh_ptailorig.so; runp new ""; done | tee -a $O/r06t_ab_ptail_waits.txt
