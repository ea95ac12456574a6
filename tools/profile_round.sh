#!/bin/bash
# Refresh the judged measurement artefacts of one round on the GPU box:  bash tools/profile_round.sh r03a [quick]
# Writes gpurun_out/<tag>_*; copy what should be judged into profiles/ afterwards.
# (counter passes carry --kernel-trace + --pmc only, one pass per counter set, as MI355X_MICROARCH.md prescribes)
# Every leg is checked: a bench leg that exits non-zero or does not end in one JSON line is REMOVED (never a 0-byte log that a
# doc could cite -- round-2 VERDICT), its stderr tail is printed, and the script exits non-zero at the end.
# `quick` = the three bench lines + the kernel-stats / PMC passes of the headline only.
TAG=${1:-rXX}
MODE=${2:-full}
export TMPDIR=/tmp
export TAG O=gpurun_out
mkdir -p $O
FAILED=""
bench_leg() {   # bench_leg <output log> <bench.py args...>
    local out=$1; shift
    timeout 900 python bench.py "$@" > "$out" 2> "$out.err"
    local rc=$?
    if [ $rc -ne 0 ] || ! tail -1 "$out" | python -c "import sys, json; json.loads(sys.stdin.read())" 2> /dev/null; then
        echo "!! FAILED LEG (rc=$rc): bench.py $*"; tail -15 "$out.err"
        mv "$out.err" "$out.FAILED.err"; rm -f "$out"
        FAILED="$FAILED [bench.py $*]"
    else
        rm -f "$out.err"
    fi
}
leg() {         # leg '<shell command>': any other step, exit status checked
    bash -o pipefail -c "$1"
    local rc=$?
    if [ $rc -ne 0 ]; then echo "!! FAILED LEG (rc=$rc): $1"; FAILED="$FAILED [${1:0:80}]"; fi
}
bench_leg $O/${TAG}_bench_feature_b32.json.log --details $O/${TAG}_bench_details.json
bench_leg $O/${TAG}_bench_train_b64.json.log --workload train --no-cpu-baseline
bench_leg $O/${TAG}_bench_predict_12cities.json.log --workload predict --steps 12 --warmup 2
bench_leg $O/${TAG}_bench_epoch.json.log --workload epoch
SRBH_TRUNK_PRECISION=f32 bench_leg $O/${TAG}_bench_feature_b32_strict_f32_trunk.json.log --steps 3 --warmup 1 --no-extras --no-cpu-baseline
bench_leg $O/${TAG}_bench_sr_train_b8.json.log --workload sr_train --steps 6 --warmup 2
SRBH_SR_BENCH_MODES=fast,mixed bench_leg $O/${TAG}_bench_sr_train_b24.json.log --workload sr_train --steps 6 --warmup 2 --batch 24
leg 'timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python bench.py --no-cpu-baseline --no-extras > $O/${TAG}_stats.log 2>&1'
leg 'cp $(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_feature_b32_kernel_stats.csv'
# (the stats CSV averages EVERY launch of a kernel, the one-tile parity forwards included: the batch-32 launches on their own)
leg 'python tools/trunk_launch_stats.py $O/${TAG}_stats > $O/${TAG}_trunk_launches_by_grid.txt'
leg 'timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_pmc_fetch.log 2>&1'
leg 'timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_pmc_write.log 2>&1'
leg 'python tools/pmc_traffic.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_hbm_traffic.json'
if [ "$MODE" != "quick" ]; then
leg 'timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/${TAG}_tr -- python bench.py --workload train --steps 6 --warmup 3 --no-extras > /dev/null 2>&1'
leg 'python tools/steady_stats.py /tmp/${TAG}_tr 4 60 --stock > $O/${TAG}_train_steady_kernel_stats.txt'
leg 'python tools/gap_stats.py /tmp/${TAG}_tr 4 12 >> $O/${TAG}_train_steady_kernel_stats.txt'
# head kernels: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, one pass each) and SQ counters
leg 'timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${TAG}_hf -- python tools/head_kernels.py 64 3 > /dev/null 2>&1'
leg 'timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${TAG}_hw -- python tools/head_kernels.py 64 3 > /dev/null 2>&1'
leg 'python tools/pmc_head.py /tmp/${TAG}_hf /tmp/${TAG}_hw $O/${TAG}_pmc_head_kernels.json 64 > /dev/null'
leg 'bash tools/pmc_sq_head.sh > /dev/null 2>&1 && cp $O/sq_head_summary.txt $O/${TAG}_sq_counters_head.txt'
leg 'SRBH_PREDICT_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/${TAG}_trp0 -- python bench.py --workload predict --steps 6 --warmup 2 > /dev/null 2>&1'
leg 'python tools/steady_stats.py /tmp/${TAG}_trp0 20 40 > $O/${TAG}_predict_steady_kernel_stats.txt'
leg 'bash tools/pmc_sq.sh > /dev/null 2>&1 && cp $O/sq/summary.txt $O/${TAG}_sq_counters_ptrunk.txt'
fi
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
find $O -name "${TAG}_*" -type f -size 0 -print -delete | sed 's/^/!! removed EMPTY artefact: /'
for f in $O/${TAG}_bench_feature_b32.json.log $O/${TAG}_bench_train_b64.json.log $O/${TAG}_bench_predict_12cities.json.log; do [ -s $f ] && tail -1 $f; done
head -4 $O/${TAG}_bench_feature_b32_kernel_stats.csv; cat $O/${TAG}_sq_counters_ptrunk.txt 2> /dev/null
if [ -n "$FAILED" ]; then echo "!! profile_round: FAILED legs:$FAILED"; exit 1; fi
echo "profile_round: all legs OK"
