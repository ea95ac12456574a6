#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_feature_h16.py tests/test_gpu_head_f16.py tests/test_gpu_model.py tests/test_gpu_graph_lifetime.py 2>&1 | tail -5
cd /tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/r04e_tr -- python $GRAFT_REPO_ROOT/bench.py --workload train --steps 6 --warmup 3 --no-extras > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/steady_stats.py /tmp/r04e_tr 4 60 --stock > $O/r04e_train_steady_kernel_stats.txt
python tools/gap_stats.py /tmp/r04e_tr 4 12 >> $O/r04e_train_steady_kernel_stats.txt
grep -A200 'stock-op kernels' $O/r04e_train_steady_kernel_stats.txt | head -120
