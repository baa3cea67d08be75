#!/bin/bash
# start / end of every kernel of the LAST decode of scripts/kernel_times.py (rocprofv3 --kernel-trace) -> gpurun_out/r3/tl_<tag>.txt
#   CAIRO_ZSTD_AMD_LIB=... [env for the library] scripts/ktimeline.sh <workload> <frames> <tag>
set -o pipefail
WL=${1:-full_4a}; N=${2:-10000}; TAG=${3:-$WL}
O=gpurun_out/${KT_OUT:-r5}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf $O/tlr_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tlr_$TAG -- python3 scripts/kernel_times.py --child $WL $N > $O/tlr_$TAG.log 2>&1 || { tail -5 $O/tlr_$TAG.log; exit 1; }
f=$(find $O/tlr_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/tl_$TAG.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("cz")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last decode = everything from the last pass-0 scan on
scans = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("cz_scan_kernel")]
first = scans[-2] if len(scans) >= 2 else 0
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f'{r["Kernel_Name"][:34]:34s} q {r.get("Queue_Id", "?"):>3s} start {s/1e6:8.3f} ms  end {e/1e6:8.3f} ms  dur {(e-s)/1e6:8.3f} ms')
PY
rm -rf $O/tlr_$TAG
