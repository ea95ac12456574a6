#!/bin/bash
# one whole SR-stage trainer iteration at batch 8 with the discriminator's 3x3 convs stock / on libsrbh (all, or the <= 64-channel ones); every variant
# once UNTIMED first (MIOpen searches its kernels per shape + layout on first use and caches them on disk), then timed twice, interleaved
export TMPDIR=/tmp O=gpurun_out
for v in stock libsrbh64 libsrbh; do SRBH_SR_DISC=$v timeout 600 python tools/sr_iteration_phases.py 8 > /dev/null 2>&1; done
(for r in 1 2; do for v in stock libsrbh64 libsrbh; do echo "SRBH_SR_DISC=$v: $(SRBH_SR_DISC=$v timeout 600 python tools/sr_iteration_phases.py 8 2>&1 | grep free-running)"; done; done) | tee $O/r05cs_sr_disc_variants.txt
