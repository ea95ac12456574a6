"""developer aid: timeline of ONE steady pipelined training step from a rocprofv3 --kernel-trace csv: where the trunk launches of the
second stream sit relative to the main stream's phases (big = head kernels, small = encoder / decoder chains), how long the main
stream waits for them.   usage: pipe_timeline.py <trace dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
a, b = adam[-3], adam[-2]                      # one step: Adam to Adam
sel = rows[a + 1:b + 1]
t0 = sel[0][0]
trunk = [(s - t0, e - t0) for s, e, k, q in sel if "ptrunk3_kernel" in k]
print("step (adam to adam): %.2f ms, %d kernels" % ((sel[-1][1] - t0) / 1e6, len(sel)))
print("trunk launches (ms from the step's start):", ["%.2f-%.2f" % (s / 1e6, e / 1e6) for s, e in trunk])
qs = collections.Counter(q for _, _, _, q in sel)
print("kernels per queue:", dict(qs))
# phases of the main stream: classify kernels by duration class and name
def cls(k):
    for n in ("hconv", "hbwd16", "hwgrad", "bn_add_relu", "bn_bwd", "ptail", "conv_first", "conv3x3_f16", "cedice", "mse", "ps2_inverse", "hpack", "bn_finalize", "hconv_up"):
        if n in k: return "head"
    if "ptrunk3" in k: return "trunk"
    return "small"
# bucket the step into 0.5 ms bins: busy time per class
binw = 500000
nb = (sel[-1][1] - t0) // binw + 1
bins = [collections.Counter() for _ in range(nb)]
for s, e, k, q in sel:
    c = cls(k)
    x = s - t0
    while x < e - t0:
        bi = x // binw
        nx = min(e - t0, (bi + 1) * binw)
        bins[bi][c] += nx - x
        x = nx
print("per 0.5 ms bin: busy us of head / small / trunk kernels (sum of durations: > 500 means overlap)")
for i, c in enumerate(bins):
    print("%5.1f ms  head %4d  small %4d  trunk %4d" % (i * 0.5, c["head"] // 1000, c["small"] // 1000, c["trunk"] // 1000))
