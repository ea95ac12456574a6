#!/bin/bash
# hconv_entry64_kernel on v_mfma_f32_16x16x32_f16 (two chunks per instruction): head suites, graph timing, predict + train A/B against the
# one-chunk-per-instruction kernel (build/variants/libsrbh_entk16.so: this tree with the old header)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; TAG=${TAG:-r06aa}
timeout 2400 python -m pytest tests/test_gpu_feature_h16.py tests/test_gpu_head.py tests/test_gpu_head_f16.py tests/test_gpu_model.py tests/test_gpu_hblock16.py tests/test_gpu_grad_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/${TAG}_tests_entry64_k32.txt 2>&1; tail -3 $O/${TAG}_tests_entry64_k32.txt
for r in 1 2; do
  echo "old: $(SRBH_LIB_PATH=build/variants/libsrbh_entk16.so python tools/predict_parts.py 256 20 2>/dev/null | grep '^batch' | tail -1)" >> $O/${TAG}_predict_parts.txt
  echo "new: $(python tools/predict_parts.py 256 20 2>/dev/null | grep '^batch' | tail -1)" >> $O/${TAG}_predict_parts.txt
done
cat $O/${TAG}_predict_parts.txt
for r in 1 2 3; do
  for v in build/variants/libsrbh_entk16.so ""; do
    x=$(SRBH_LIB_PATH=$v timeout 900 python bench.py --workload predict --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('ms_per_step'))")
    echo "predict ${v:-new} $x" >> $O/${TAG}_ab_entry64.txt
    x=$(SRBH_LIB_PATH=$v timeout 900 python bench.py --workload train --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('ms_per_step'))")
    echo "train ${v:-new} $x" >> $O/${TAG}_ab_entry64.txt
  done
done
cat $O/${TAG}_ab_entry64.txt
