#!/bin/bash
# kernel durations of ONE large frame of the mix decoded alone with the pre-pass and the literals pass (rank by decoded size: $1)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r2; rm -rf gpurun_out/r2/kt_single
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2/kt_single -- python3 scripts/mix_single_profile.py ${1:-0} lit > gpurun_out/r2/kt_single.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2/kt_single/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('cz_')]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last_scan = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("cz_scan_kernel") and (i == 0 or not rows[i - 1]["Kernel_Name"].startswith("cz_scan_kernel")))
t0=int(rows[last_scan]['Start_Timestamp'])
for r in rows[last_scan:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{r['Kernel_Name'][:28]:28s} start {(s-t0)/1e6:8.3f} ms  dur {(e-s)/1e6:8.3f} ms")
PY
rm -rf gpurun_out/r2/kt_single
