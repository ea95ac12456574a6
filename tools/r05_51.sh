#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_sidework.py tests/test_gpu_head.py -x -q 2>&1 | tail -3
python tools/host_bound_probe.py 2>&1 | grep "batch 2" | tee $O/r05az_host_after_stream_ptr.txt
run() { echo "$* $(env "$@" timeout 600 python bench.py --workload train --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")"; }
for r in 1 2 3; do
run SRBH_TRAIN_PIPELINE=1
run SRBH_TRAIN_PIPELINE=1 SRBH_HOST_SPIN_US=2
run SRBH_TRAIN_PIPELINE=1 SRBH_HOST_SPIN_US=4
done 2>&1 | tee -a $O/r05az_host_after_stream_ptr.txt
