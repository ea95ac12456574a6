#!/bin/bash
export TMPDIR=/tmp O=gpurun_out
bash tools/prof_variants.sh 2>&1 | tee $O/r05w_trunk_timeline.txt
