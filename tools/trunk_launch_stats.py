"""per-grid duration statistics of the trunk kernel from a rocprofv3 kernel trace (tools/profile_round.sh): the bench process also runs
one-tile parity forwards, whose launches must not be averaged with the batch-32 ones (round-5 VERDICT, measurement hygiene).
usage: trunk_launch_stats.py <rocprofv3 output dir> [kernel substring]"""
import csv, glob, os, sys
d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "ptrunk3_kernel")
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    sys.exit("no kernel trace under " + d)
groups = {}
for f in files:
    for r in csv.DictReader(open(f)):
        if pat not in r.get("Kernel_Name", ""):
            continue
        wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        groups.setdefault(wgs, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for wgs, v in sorted(groups.items()):
    v.sort()
    print(f"{pat}: {wgs:4d} workgroups ({wgs // 8} tiles of 64x64)  launches {len(v):4d}  avg {sum(v) / len(v):.4f} ms  median {v[len(v) // 2]:.4f}  min {v[0]:.4f}  max {v[-1]:.4f}")
