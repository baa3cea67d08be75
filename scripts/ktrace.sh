#!/bin/bash
# per-kernel durations of one workload: rocprofv3 --kernel-trace --stats over scripts/kernel_times.py --child -> gpurun_out/r3/kt_<workload>.txt
set -o pipefail
WL=${1:-full_4a}; N=${2:-10000}
O=gpurun_out/${KT_OUT:-r4}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf $O/ktr_$WL
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktr_$WL -- python3 scripts/kernel_times.py --child $WL $N > $O/ktr_$WL.log 2>&1 || { tail -5 $O/ktr_$WL.log; exit 1; }
f=$(find $O/ktr_$WL -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/kt_$WL.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("cz_") or r["Name"].startswith("czx"):
        print(f'{r["Name"][:40]:40s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}')
PY
rm -rf $O/ktr_$WL
