"""developer aid: durations of the trunk launches in the order they ran, from a rocprofv3 --kernel-trace csv of `bench.py --workload train`
(is the first launch of a step slower than the second?)   usage: trunk_in_step.py <trace dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
d = [(e - s) / 1e6 for s, e, k in rows if "ptrunk3_kernel" in k]
print("ptrunk3 launches (ms), last 16:", " ".join("%.3f" % v for v in d[-16:]))
prev = None
gaps = []
for s, e, k in rows:
    if "ptrunk3_kernel" in k and prev is not None:
        gaps.append(((s - prev[1]) / 1e3, prev[2][:40]))
    prev = (s, e, k)
print("kernel right before each trunk launch + gap (us), last 8:", gaps[-8:])
