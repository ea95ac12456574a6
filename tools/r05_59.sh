#!/bin/bash
# the fixed poll window (masked weight statement behind the check): pipelined soak, every 2nd prefetch compared, two runs + parity tests
export TMPDIR=/tmp O=gpurun_out
(
echo "== HEAD (poll window holds unmasked statements only)"; timeout 900 python tools/soak_pipelined.py 1500 2
echo "== again"; timeout 900 python tools/soak_pipelined.py 1500 2
echo "== standalone soak"; timeout 600 python tools/soak.py 400
) 2>&1 | grep -v amdgpu.ids | tee $O/r05bp_soak_fixed.txt
timeout 1500 python -m pytest tests/test_gpu_rrdbnet.py tests/test_gpu_feature_h16.py tests/test_sr_stage.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5 | tee $O/r05bp_tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $O/r05bp_bench.json
