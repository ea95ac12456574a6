#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_tr -- python bench.py --workload train --steps 6 --warmup 3 --no-extras --no-cpu-baseline > /dev/null 2>&1
python tools/pipe_timeline.py /tmp/pt_tr | tee $O/r05af_pipe_timeline.txt
