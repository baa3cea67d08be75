#!/bin/bash
# SQ instruction counters of one workload's kernels (one rocprofv3 --pmc pass, no trace domains) -> gpurun_out/r2/sq_<workload>.txt
# usage: sq_counters.sh [workload] [frames]
set -o pipefail
WL=${1:-full_4a}; N=${2:-10000}
O=gpurun_out/${SQ_OUT:-r5}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf $O/sq_$WL
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_$WL -- python3 scripts/kernel_times.py --child $WL $N > $O/sq_$WL.log 2>&1 || { tail -5 $O/sq_$WL.log; exit 1; }
python3 scripts/sq_summary.py "$O/sq_$WL" $O/sq_$WL.json $WL $N > $O/sq_$WL.txt
rm -rf $O/sq_$WL
cat $O/sq_$WL.txt
