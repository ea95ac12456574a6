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
# what ran in the 6 ms before the first trunk launch of the last step: total time per kernel family
idx = [i for i, (s, e, k) in enumerate(rows) if "ptrunk3_kernel" in k]
if len(idx) >= 2:
    first = idx[-2]
    t0 = rows[first][0]
    fam = {}
    for s, e, k in rows[:first]:
        if e >= t0 - 6_000_000:
            name = k.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:48]
            fam[name] = fam.get(name, 0.0) + (e - s) / 1e3
    print("kernel time (us) in the 6 ms before the step's first trunk launch, by kernel:")
    for name, us in sorted(fam.items(), key=lambda kv: -kv[1])[:12]:
        print("  %8.1f  %s" % (us, name))
    print("the 10 kernels right before it:", [rows[i][2].split("(")[0].replace("void ", "")[:32] for i in range(max(0, first - 10), first)])
