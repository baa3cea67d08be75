#!/usr/bin/env python3
"""Per-launch SQ counter totals of the LAST step in a rocprofv3 --pmc csv directory (see scripts/sq_counters.sh).  A kernel
launched more than once per step is numbered in launch order, as in scripts/pmc_summary.py."""
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
rows = [(k, by[(i, k)]) for (i, k) in sorted(by)]
has_scan = any(k == "cz_scan_kernel" for k, _ in rows)
steps, cur, prev = [], [], None
for k, e in rows:
    if (has_scan and k == "cz_scan_kernel" and prev != "cz_scan_kernel") or not has_scan:
        if cur:
            steps.append(cur)
        cur = []
    cur.append((k, e))
    prev = k
if cur:
    steps.append(cur)
last = steps[-1]
names = [k for k, _ in last]
seen = {}
for k, e in last:
    seen[k] = seen.get(k, 0) + 1
    print(k + (f"#{seen[k]}" if names.count(k) > 1 else ""), " ".join(f"{c}={e[c]:.5g}" for c in sorted(e)))
