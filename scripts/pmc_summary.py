#!/usr/bin/env python3
"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) of bench.py into
per-kernel HBM traffic per launch.  usage: pmc_summary.py <fetch_dir> <write_dir> <bench_json> <out_json>"""
import csv
import glob
import json
import sys


def per_dispatch(d, counter):
    """[(dispatch id, kernel name, value summed over the XCD rows)] of the cz_ kernels, in dispatch order."""
    by = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"].split("(")[0]
            if not k.startswith("cz_"):
                continue
            e = by.setdefault(int(row["Dispatch_Id"]), [k, 0.0])
            e[1] += float(row["Counter_Value"])
    return [(i, by[i][0], by[i][1]) for i in sorted(by)]


def per_step(rows):
    """Groups the dispatches into steps (a step starts at its first cz_scan_kernel, or at every dispatch when there is no
    pre-pass) and returns the mean KB per step for each launch of the step, named by kernel and position."""
    steps, cur = [], []
    has_scan = any(k == "cz_scan_kernel" for _, k, _ in rows)
    prev = None
    for _, k, v in rows:
        if (has_scan and k == "cz_scan_kernel" and prev != "cz_scan_kernel") or not has_scan:
            if cur:
                steps.append(cur)
            cur = []
        cur.append((k, v))
        prev = k
    if cur:
        steps.append(cur)
    shape = [k for k, _ in steps[-1]]
    steps = [s for s in steps if [k for k, _ in s] == shape]
    names, seen = [], {}
    for k in shape:
        seen[k] = seen.get(k, 0) + 1
        names.append(k if shape.count(k) == 1 else f"{k}#{seen[k]}")
    return {n: sum(s[i][1] for s in steps) / len(steps) for i, n in enumerate(names)}, len(steps)


def main():
    fd, wd, bj, oj = sys.argv[1:5]
    (fetch, nf), (write, nw) = per_step(per_dispatch(fd, "FETCH_SIZE")), per_step(per_dispatch(wd, "WRITE_SIZE"))
    line = json.load(open(bj))
    cfg = line["config"]
    res = {"workload": cfg["workload"], "frames": cfg["frames_per_gpu"], "steps_averaged": [nf, nw],
           "launch_naming": "a kernel launched more than once per step is numbered in launch order (cz_scan_kernel#1 counts, #2 places)",
           "per_kernel": {},
           "algorithmic_read_bytes": cfg.get("compressed_bytes_rank0", cfg.get("compressed_bytes_per_gpu")),
           "algorithmic_write_bytes": cfg.get("decoded_bytes_rank0", cfg.get("decoded_bytes_per_gpu")),
           # bench.py quotes this file only for the kernel sources and launch options it was measured on
           "kernel_source_hash": line.get("kernel_source_hash"), "chain_prepass": line.get("chain_prepass"), "exec_kernel": bool(line.get("exec_kernel")), "wexec_kernel": bool(line.get("wexec_kernel", False))}
    for k in sorted(set(fetch) | set(write)):
        res["per_kernel"][k] = {"FETCH_SIZE_KB_per_launch": fetch.get(k), "WRITE_SIZE_KB_per_launch": write.get(k)}
    res["fetch_bytes_uncorrected"] = sum(fetch.values()) * 1024.0
    res["write_bytes"] = sum(write.values()) * 1024.0
    res["fetch_over_algorithmic_read"] = res["fetch_bytes_uncorrected"] / res["algorithmic_read_bytes"]
    res["write_over_algorithmic_write"] = res["write_bytes"] / res["algorithmic_write_bytes"]
    res["fetch_bytes_if_all_reads_were_wide"] = 2.0 * res["fetch_bytes_uncorrected"]
    res["note"] = ("FETCH_SIZE / WRITE_SIZE are in KB, counted at the L2's fabric side (Infinity-Cache hits included). On gfx950 FETCH_SIZE "
                   "reports half the bytes of a 16 B/lane coalesced read (MI355X_MICROARCH.md, HBM section): the true figure lies between "
                   "fetch_bytes_uncorrected (all reads narrow) and twice that (all reads wide); the decoder's reads are mostly narrow "
                   "(match sources, records), its input staging is wide")
    json.dump(res, open(oj, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
