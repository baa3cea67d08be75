#!/bin/bash
# Any set of counters for one workload's kernels (one rocprofv3 --pmc pass per call, no trace domains) -> gpurun_out/r4/pmc_<tag>.txt
#   scripts/pmc_any.sh <tag> <workload> <frames> COUNTER [COUNTER ...]      (rocprofv3 -L lists them; totals of the last step per kernel)
set -o pipefail
TAG=$1; WL=$2; N=$3; shift; shift; shift
O=gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf $O/pmcd_$TAG
timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $O/pmcd_$TAG -- python3 scripts/kernel_times.py --child $WL $N > $O/pmc_$TAG.log 2>&1 || { tail -5 $O/pmc_$TAG.log; exit 1; }
python3 scripts/sq_summary.py "$O/pmcd_$TAG" > $O/pmc_$TAG.txt
rm -rf $O/pmcd_$TAG
grep "execute_frames_kernel\|wexec\|chain_kernel\|huf_kernel" $O/pmc_$TAG.txt
