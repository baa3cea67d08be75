#!/usr/bin/env python3
"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) of bench.py into
per-kernel HBM traffic per launch.  usage: pmc_summary.py <fetch_dir> <write_dir> <bench_json> <out_json>"""
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"].split("(")[0]
            if not k.startswith("cz_"):
                continue
            e = out.setdefault(k, {"sum": 0.0, "dispatches": set()})
            e["sum"] += float(row["Counter_Value"])
            e["dispatches"].add(row["Dispatch_Id"])
    return {k: v["sum"] / max(len(v["dispatches"]), 1) for k, v in out.items()}


def main():
    fd, wd, bj, oj = sys.argv[1:5]
    fetch, write = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    line = json.load(open(bj))
    cfg = line["config"]
    res = {"workload": cfg["workload"], "frames": cfg["frames_per_gpu"], "per_kernel": {},
           "algorithmic_read_bytes": cfg.get("compressed_bytes_rank0", cfg.get("compressed_bytes_per_gpu")),
           "algorithmic_write_bytes": cfg.get("decoded_bytes_rank0", cfg.get("decoded_bytes_per_gpu")),
           # bench.py quotes this file only for the kernel sources and launch options it was measured on
           "kernel_source_hash": line.get("kernel_source_hash"), "chain_prepass": line.get("chain_prepass"), "exec_kernel": bool(line.get("exec_kernel"))}
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
