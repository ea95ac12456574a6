"""developer aid: per-(kernel, grid) average duration of the 1x1-convolution kernels from a rocprofv3 kernel trace   usage: pw_trace.py <dir> [pattern]"""
import csv, glob, collections, sys
pat = sys.argv[2] if len(sys.argv) > 2 else "pw_"
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48], int(r.get("Grid_Size_X", r.get("Grid_Size", 0))), int(r.get("Workgroup_Size_X", 0)))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print("total", round(tot / 1e3), "us over the run")
for (k, g, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{k:50s} grid {g:8d} wg {w:5d} x{len(v):4d}  {sum(v) / len(v) / 1e3:7.1f} us")
