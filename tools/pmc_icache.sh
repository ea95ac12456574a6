#!/bin/bash
# developer aid: instruction-cache behaviour of the trunk kernel (kernel-trace + pmc only)
export TMPDIR=/tmp
OUT=gpurun_out/ic
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p1.log 2>&1
echo "rc=$?"
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ptrunk" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print("%-32s %16.0f  (n=%d)" % (k, sum(agg[k]) / len(agg[k]), len(agg[k])))
PY
tail -3 $OUT/p1.log
