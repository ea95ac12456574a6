#!/bin/bash
# hblock16_kernel on v_mfma_f32_16x16x32_f16 (two taps per instruction): tests, kernel time, predict A/B against the one-tap-per-instruction kernel
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; TAG=${TAG:-r06z}
timeout 1500 python -m pytest tests/test_gpu_hblock16.py tests/test_gpu_feature_h16.py tests/test_gpu_model.py -x -q -m gpu > $O/${TAG}_tests_hblock16_k32.txt 2>&1; tail -3 $O/${TAG}_tests_hblock16_k32.txt
for r in 1 2; do
  echo "== one tap per instruction (16x16x16)" >> $O/${TAG}_time_hblock16.txt; SRBH_LIB_PATH=build/variants/libsrbh_hb16k16.so python tools/time_hblock16.py 256 >> $O/${TAG}_time_hblock16.txt 2>&1
  echo "== two taps per instruction (16x16x32)" >> $O/${TAG}_time_hblock16.txt; python tools/time_hblock16.py 256 >> $O/${TAG}_time_hblock16.txt 2>&1
done
cat $O/${TAG}_time_hblock16.txt
for r in 1 2 3; do
  for v in build/variants/libsrbh_hb16k16.so ""; do
    x=$(SRBH_LIB_PATH=$v timeout 900 python bench.py --workload predict --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('ms_per_step'))")
    echo "${v:-new} $x" >> $O/${TAG}_ab_predict_hblock16.txt
  done
done
cat $O/${TAG}_ab_predict_hblock16.txt
