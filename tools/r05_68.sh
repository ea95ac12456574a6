#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 1500 python -m pytest tests/test_sr_stage.py tests/test_gpu_rrdbnet.py tests/test_gpu_feature_h16.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python tools/sr_iteration_phases.py 8 2>&1 | grep -v amdgpu.ids | tee $O/r05cf_sr_iteration_phases.txt
