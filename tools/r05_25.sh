#!/bin/bash
# what the staging costs inside the steps as they stand: timelines of the ablated builds (WRONG results: 1 no weight DMA, 2 no input DMA, 4 no register staging)
export TMPDIR=/tmp O=gpurun_out
bash tools/prof_variants.sh abl1 abl2 abl4 abl7 2>&1 | tee $O/r05y_trunk_ablation_timeline.txt
