#!/usr/bin/env python3
"""Per-kernel SQ counter totals of the LAST step in a rocprofv3 --pmc csv directory (see scripts/sq_counters.sh)."""
import csv
import glob
import sys

by = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("cz_"):
            continue
        e = by.setdefault((int(r["Dispatch_Id"]), k), {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
runs = {}
for (i, k) in sorted(by):
    runs.setdefault(k, []).append(by[(i, k)])
for k, rs in runs.items():
    per_step = 2 if k == "cz_scan_kernel" or (k == "cz_decode_frames_kernel" and "cz_chain_kernel" in runs) else 1
    for j, e in enumerate(rs[-per_step:]):
        print(k + (f"#{j + 1}" if per_step > 1 else ""), " ".join(f"{c}={e[c]:.5g}" for c in sorted(e)))
