"""developer aid: where the GPU idles inside a step.  From a rocprofv3 --kernel-trace csv, over the last n steps (see
steady_stats.py for the step boundary), aggregate the idle time BEFORE each kernel (start - max end so far) by kernel name.
usage: gap_stats.py <dir> <n_steps> [top]"""
import csv, glob, sys, collections
d, n = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
bounds = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]                   # a training step ends with libsrbh's one Adam launch
if len(bounds) < 2:                                                                  # (other workloads: one conv_first per RRDBNet forward -- the pipelined training step has several)
    bounds = [i for i, r in enumerate(rows) if "conv_first_kernel" in r[2]]
start = bounds[-n - 1] if len(bounds) > n else bounds[0]
end = bounds[-1]
sel = rows[start:end]
steps = max(1, len([b for b in bounds if start <= b < end]))
agg = collections.defaultdict(lambda: [0, 0, 0])
hi = sel[0][1]
busy = 0
hist = collections.Counter()
for s, e, k in sel[1:]:
    g = max(0, s - hi)
    a = agg[k.replace("(anonymous namespace)::", "")[:100]]
    a[0] += 1; a[1] += g; a[2] = max(a[2], g)
    hist[min(7, g // 10000)] += 1
    hi = max(hi, e)
tot = sum(v[1] for v in agg.values())
print("steps %d | idle %.2f ms/step of wall %.2f ms/step" % (steps, tot / steps / 1e6, (sel[-1][1] - sel[0][0]) / steps / 1e6))
print("gap histogram (10 us bins, last = >=70 us), per step:", [hist[i] // steps for i in range(8)])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%7.3f ms/step idle before | %5d calls/step | max %7.1f us | %s" % (v[1] / steps / 1e6, v[0] // steps, v[2] / 1e3, k))
