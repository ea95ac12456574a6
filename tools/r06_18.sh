# round 6: the RRDB-closing epilogue's stream loads as asm loads with manual counted waits: trunk tests (bit-identity with the per-layer path, parity),
# same-box A/B against the previous header (build/variants/libsrbh_r2orig.so), per-RDB timeline of both
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rrdbnet.py tests/test_gpu_feature_h16.py tests/test_gpu_parity_sweep.py tests/test_gpu_predict_sharded.py tests/test_sr_stage.py -q -x -m gpu 2>&1 | tail -3 | tee $O/r06s_tests_trunk.txt
bash tools/ab_variants.sh r2orig 2>&1 | tee $O/r06s_ab_r2_waits.txt
bash tools/ab_variants.sh r2orig 2>&1 | tee -a $O/r06s_ab_r2_waits.txt
for v in new orig; do L=""; [ $v = orig ] && L=build/variants/libsrbh_r2orig.so; SRBH_LIB_PATH=$L SRBH_PT_PROF=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "\[srbh\]" | tail -8 | sed "s/^/$v /"; done | tee $O/r06s_trunk_timeline.txt
