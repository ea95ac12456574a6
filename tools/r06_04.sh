# round 6: gradient-parity test + the train bench line carrying `parity`; the MFMA ceiling of this package re-measured at HEAD with an SMI trace
export TMPDIR=/tmp O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_grad_parity.py tests/test_gpu_pipeline.py tests/test_gpu_adam.py tests/test_gpu_dp_trainstep.py -q 2>&1 | tail -5 | tee $O/r06d_gpu_tests_new.txt
timeout 900 python bench.py --workload train --steps 20 --warmup 8 > $O/r06d_bench_train_b64.json.log 2> $O/r06d_bench_train.err; tail -c 2500 $O/r06d_bench_train_b64.json.log; tail -3 $O/r06d_bench_train.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o /tmp/mfma_ceiling 2>/dev/null
python tools/dump_trunk_operands.py $O/trunk_acts.bin $O/trunk_weights.bin
( while true; do echo "t=$(date +%s.%N)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk'; sleep 0.5; done ) > $O/r06d_mfma_ceiling_smi.log 2>&1 &
SMI=$!
CEIL_SECONDS=4 CEIL_SKIP_ORDER=1 timeout 600 /tmp/mfma_ceiling $O/trunk_acts.bin $O/trunk_weights.bin 2>&1 | while IFS= read -r line; do echo "t=$(date +%s.%N) $line"; done | tee $O/r06d_mfma_ceiling.txt
kill $SMI
rm -f $O/trunk_acts.bin $O/trunk_weights.bin
