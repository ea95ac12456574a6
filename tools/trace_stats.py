"""developer aid: per-kernel totals of a rocprofv3 --kernel-trace csv, skipping the first `skip` fraction of the launches
(warm-up: MIOpen's solver search).  usage: trace_stats.py <dir> [skip=0.33] [top=30]"""
import csv, glob, sys, collections
d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.33
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * skip):]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in rows:
    a = agg[k.replace("(anonymous namespace)::", "")[:110]]
    a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
print("kernel time %.1f ms in %d launches (after skipping the first %.0f %%)" % (tot / 1e6, len(rows), skip * 100))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%6.1f %% %7d calls %9.1f us avg  %s" % (100.0 * v[1] / tot, v[0], v[1] / v[0] / 1e3, k))
