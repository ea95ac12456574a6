#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
for v in 0 1; do
  SRBH_HRFEAT_FIRST=$v timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$v -- python bench.py --workload train --steps 6 --warmup 3 --no-extras --no-cpu-baseline > /dev/null 2>&1
  echo "== SRBH_HRFEAT_FIRST=$v"; python tools/trunk_in_step.py /tmp/tr$v
done 2>&1 | tee $O/r05n_trunk_in_step.txt
python tools/tiny_clock_probe.py 2>&1 | grep -v amdgpu | tee $O/r05n_tiny_clock_probe.txt
