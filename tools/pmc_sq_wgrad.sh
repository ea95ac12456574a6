#!/bin/bash
# developer aid: SQ counter passes over srbh_trunk_wgrad alone (tools/time_trunk_wgrad.py; kernel-trace + pmc only, one pass per counter set)
export TMPDIR=/tmp
B=${1:-24}
OUT=gpurun_out/sq_wgrad
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python tools/time_trunk_wgrad.py $B > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/sq_wgrad/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trunk_wgrad_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/sq_wgrad/summary.txt", "w") as o:
    for k in sorted(agg):
        line = "%-32s %16.0f  (n=%d)" % (k, sum(agg[k]) / len(agg[k]), len(agg[k]))
        print(line); o.write(line + "\n")
PY
find $OUT -name "*.csv" -size +2M -delete
