#!/usr/bin/env python3
"""Per-launch SQ counter totals of the LAST step in a rocprofv3 --pmc csv directory (see scripts/sq_counters.sh).  A kernel
launched more than once per step is numbered in launch order, as in scripts/pmc_summary.py.
    sq_summary.py <csv dir> [<out.json> <workload> <frames>]
With an output path the totals are also written as JSON, stamped with the hash of the kernel sources (bench.py quotes
roofline.issue_bound_ms from that file only for the code it was measured on)."""
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

if len(sys.argv) > 2:
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    tot = {}
    per = {}
    seen = {}
    for k, e in last:
        seen[k] = seen.get(k, 0) + 1
        per[k + (f"#{seen[k]}" if names.count(k) > 1 else "")] = {c: e[c] for c in sorted(e)}
        for c, v in e.items():
            tot[c] = tot.get(c, 0.0) + v
    json.dump({"kernel_source_hash": bench._kernel_source_hash(), "workload": sys.argv[3] if len(sys.argv) > 3 else None, "frames": int(sys.argv[4]) if len(sys.argv) > 4 else None,
               "what": "SQ instruction counters of one step (rocprofv3 --pmc pass of scripts/kernel_times.py --child: the kernels of a step run one after the other there)",
               "per_kernel": per, "step_totals": tot}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
