#!/bin/bash
# LDS-tiled 1x1-conv GEMM: tests, per-shape time against pw_gemm_kernel, predict parts
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_dwconv.py tests/test_gpu_mbconv.py -x -q -m gpu > $O/${TAG:-r06v}_tests_pwlds.txt 2>&1; tail -3 $O/${TAG:-r06v}_tests_pwlds.txt
for r in 1 2; do
  echo "== pw_gemm_kernel (SRBH_PW_LDS=0)" >> $O/${TAG:-r06v}_time_pwconv.txt; SRBH_PW_LDS=0 python tools/time_pwconv.py 256 >> $O/${TAG:-r06v}_time_pwconv.txt 2>&1
  echo "== LDS-tiled" >> $O/${TAG:-r06v}_time_pwconv.txt; python tools/time_pwconv.py 256 >> $O/${TAG:-r06v}_time_pwconv.txt 2>&1
done
grep "all 1x1" $O/${TAG:-r06v}_time_pwconv.txt
for r in 1 2; do
  echo "pw_gemm_kernel: $(SRBH_PW_LDS=0 python tools/predict_parts.py 256 20 | tail -1)" >> $O/${TAG:-r06v}_predict_parts.txt
  echo "LDS-tiled:      $(python tools/predict_parts.py 256 20 | tail -1)" >> $O/${TAG:-r06v}_predict_parts.txt
done
cat $O/${TAG:-r06v}_predict_parts.txt
