#!/bin/bash
# rocprofv3 kernel stats of whole SR-stage trainer iterations (batch 8, fast mode); one untimed run first (MIOpen's first-use kernel search)
export TMPDIR=/tmp O=gpurun_out
timeout 600 python tools/sr_iteration_phases.py 8 > /dev/null 2>&1
rm -rf /tmp/sri
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sri -- python tools/sr_iteration_phases.py 8 > /tmp/sri.log 2>&1
f=$(find /tmp/sri -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/r05cx_sr_iteration_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# tools/sr_iteration_phases.py 8 under rocprofv3 --kernel-trace --stats: 12 iterations in all (2 + 5 phase-timed + 5 free-running); kernel time {tot / 12 / 1e6:.2f} ms per iteration")
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs']) / 12 / 1e6:8.3f} ms/iter  {float(r['Percentage']):5.1f} %  {int(r['Calls']) / 12:7.1f} calls/iter  {r['Name'][:150]}")
PY
