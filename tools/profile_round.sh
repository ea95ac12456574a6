#!/bin/bash
# Refresh the judged measurement artefacts of one round on the GPU box:  bash tools/profile_round.sh r01e
# Writes gpurun_out/<tag>_*; copy what should be judged into profiles/ afterwards.
# (counter passes carry --kernel-trace + --pmc only, one pass per counter set, as MI355X_MICROARCH.md prescribes)
TAG=${1:-rXX}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python bench.py > $O/${TAG}_bench_feature_b32.json.log 2> $O/${TAG}_bench.err
timeout 600 python bench.py --workload train --no-cpu-baseline > $O/${TAG}_bench_train_b64.json.log 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --workload predict --steps 12 --warmup 2 > $O/${TAG}_bench_predict_12cities.json.log 2>> $O/${TAG}_bench.err
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/${TAG}_tr -- python bench.py --workload train --steps 6 --warmup 3 --no-extras > /dev/null 2>&1
python tools/steady_stats.py /tmp/${TAG}_tr 4 40 > $O/${TAG}_train_steady_kernel_stats.txt
python tools/gap_stats.py /tmp/${TAG}_tr 4 12 >> $O/${TAG}_train_steady_kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python bench.py --no-cpu-baseline --no-extras > $O/${TAG}_stats.log 2>&1
cp $(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_feature_b32_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_pmc_write.log 2>&1
python tools/pmc_traffic.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_hbm_traffic.json
# head kernels: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, one pass each) and SQ counters
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${TAG}_hf -- python tools/head_kernels.py 64 3 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${TAG}_hw -- python tools/head_kernels.py 64 3 > /dev/null 2>&1
python tools/pmc_head.py /tmp/${TAG}_hf /tmp/${TAG}_hw $O/${TAG}_pmc_head_kernels.json 64 > /dev/null
bash tools/pmc_sq_head.sh > /dev/null 2>&1
cp $O/sq_head_summary.txt $O/${TAG}_sq_counters_head.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/${TAG}_trp -- python bench.py --workload predict --steps 6 --warmup 2 > /dev/null 2>&1
SRBH_PREDICT_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/${TAG}_trp0 -- python bench.py --workload predict --steps 6 --warmup 2 > /dev/null 2>&1
python tools/steady_stats.py /tmp/${TAG}_trp0 20 40 > $O/${TAG}_predict_steady_kernel_stats.txt
bash tools/pmc_sq.sh > /dev/null 2>&1
cp $O/sq/summary.txt $O/${TAG}_sq_counters_ptrunk.txt
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
tail -1 $O/${TAG}_bench_feature_b32.json.log; tail -1 $O/${TAG}_bench_train_b64.json.log; head -4 $O/${TAG}_bench_feature_b32_kernel_stats.csv; cat $O/${TAG}_sq_counters_ptrunk.txt
