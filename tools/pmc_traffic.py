#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<tag>_pmc_hbm_traffic.json.

Usage (on the GPU box, two separate passes as MI355X_MICROARCH.md prescribes):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_hbm_traffic.json
Units: rocprofv3 reports both counters in KiB; FETCH_SIZE is doubled (the guide's gfx950 correction: 64 B are tallied
per 128 B request on wide coalesced reads).
"""
import csv, glob, json, os, sys, collections


def load(d, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            per["%s grid=%s" % (name, r["Grid_Size"])].append(float(r["Counter_Value"]))
    return per


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    kernels = {}
    tot = 0.0
    launches_ref = None
    for k in sorted(set(fe) | set(wr)):
        if not ("srbh" in k or "ptrunk" in k or "ptail" in k or "conv_first" in k):
            continue
        e = {"launches": max(len(fe.get(k, [])), len(wr.get(k, [])))}
        if k in fe:
            e["FETCH_SIZE_KiB_avg_per_launch"] = sum(fe[k]) / len(fe[k])
        if k in wr:
            e["WRITE_SIZE_KiB_avg_per_launch"] = sum(wr[k]) / len(wr[k])
        kernels[k] = e
        if "ptrunk" in k and "reset" not in k:
            launches_ref = max(launches_ref or 0, e["launches"])    # (the bench's one-tile parity forwards add a second, small grid)
    forwards = launches_ref or 1
    for k, e in kernels.items():
        per_fwd = e["launches"] / forwards
        tot += per_fwd * (2.0 * e.get("FETCH_SIZE_KiB_avg_per_launch", 0.0) + e.get("WRITE_SIZE_KiB_avg_per_launch", 0.0)) * 1024.0
    ptr = sorted((e for k, e in kernels.items() if "ptrunk" in k and "reset" not in k), key=lambda e: -e["launches"])
    res = {
        "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (separate passes)",
        "units": "KiB as reported; FETCH_SIZE doubled in the totals (MI355X_MICROARCH.md #HBM)",
        "forwards_in_run": forwards,
        "kernels": kernels,
        "hbm_bytes_per_forward": tot,
    }
    if ptr:
        res["dominant_kernel_hbm_bytes_per_launch"] = (2.0 * ptr[0].get("FETCH_SIZE_KiB_avg_per_launch", 0.0) + ptr[0].get("WRITE_SIZE_KiB_avg_per_launch", 0.0)) * 1024.0
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in res if k != "kernels"}, indent=1))


if __name__ == "__main__":
    main()
