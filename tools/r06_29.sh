#!/bin/bash
# ptail_pipe_kernel (MFMAs, epilogue stores and input DMA of consecutive tiles overlapped) against the serial ptail_kernel (SRBH_PTAIL_PIPE=0)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; TAG=${TAG:-r06ah}
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_rrdbnet.py tests/test_gpu_feature_h16.py tests/test_gpu_parity_sweep.py tests/test_gpu_predict_sharded.py -x -q -m gpu > $O/${TAG}_tests_ptail_pipe.txt 2>&1; tail -3 $O/${TAG}_tests_ptail_pipe.txt
run() { SRBH_PTAIL_PIPE=$2 timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], 'outside the trunk:', round(d['ms_per_step']-d['roofline']['avg_launch_ms'],4), 'parity', d['parity']['vs_strict_f32_gpu_path']['rel_l2'])"; }
for r in 1 2 3; do run serial 0; run pipelined 1; done | tee $O/${TAG}_ab_ptail_pipe.txt
for r in 1 2; do
  for v in 0 1; do
    x=$(SRBH_PTAIL_PIPE=$v timeout 900 python bench.py --workload predict --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d.get('ms_per_step'))")
    echo "predict pipe=$v $x" | tee -a $O/${TAG}_ab_ptail_pipe.txt
  done
done
