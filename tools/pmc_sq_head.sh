#!/bin/bash
# developer aid: SQ counter passes over tools/head_kernels.py (kernel-trace + pmc only, one pass per counter set); per-kernel averages
export TMPDIR=/tmp
OUT=/tmp/sqh
mkdir -p $OUT gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_VMEM SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python tools/head_kernels.py 64 2 > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/sqh/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if any(p in n for p in ("hconv", "hwgrad", "bn_bwd", "bn_add_relu")):
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/sq_head_summary.txt", "w") as o:
    for n in sorted(agg):
        o.write(n + "\n"); print(n)
        for k in sorted(agg[n]):
            line = "   %-32s %16.0f" % (k, sum(agg[n][k]) / len(agg[n][k]))
            print(line); o.write(line + "\n")
PY
