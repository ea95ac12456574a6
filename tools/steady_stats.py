"""developer aid: per-kernel time of the LAST n steps of a bench run from a rocprofv3 --kernel-trace csv (the first steps
carry MIOpen's solver search, which runs every applicable kernel incl. the naive ones).  A step boundary = a launch of
ptrunk_kernel that follows a non-ptrunk kernel.   usage: steady_stats.py <dir with *kernel_trace.csv> <n_steps> [top] [calls]   (calls: sort by launch count)"""
import csv, glob, sys, collections
d, n = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
bounds = [i for i, r in enumerate(rows) if "ptrunk" in r[2] and "_kernel" in r[2] and (i == 0 or "ptrunk" not in rows[i - 1][2])]
start = bounds[-n - 1] if len(bounds) > n else bounds[0]
end = bounds[-1]
sel = rows[start:end]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in sel:
    a = agg[k.replace("(anonymous namespace)::", "")[:110]]
    a[0] += 1; a[1] += e - s
steps = max(1, len([b for b in bounds if start <= b < end]))
tot = sum(v[1] for v in agg.values())
wall = (sel[-1][1] - sel[0][0]) if sel else 0
print("steps %d | kernel time %.2f ms/step | wall %.2f ms/step | %d launches/step" % (steps, tot / steps / 1e6, wall / steps / 1e6, len(sel) // steps))
bycalls = len(sys.argv) > 4
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0 if bycalls else 1])[:top]:
    print("%7.3f ms/step %5.1f %% %6d calls/step  %s" % (v[1] / steps / 1e6, 100.0 * v[1] / tot, v[0] // steps, k))
