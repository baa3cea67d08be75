#!/bin/bash
# Wave-cycle breakdown of one workload's kernels: two rocprofv3 --pmc passes (no trace domains) with the SQ wait / active / LDS counters
#   -> gpurun_out/<round>/sq_stalls_<workload>.txt.   usage: sq_stalls.sh [workload] [frames]      (SQ_OUT=<round>, default r5)
# Reading: SQ_WAVE_CYCLES = sum over waves of resident cycles (x4: counted every 4 clocks); SQ_WAIT_ANY = of those, waiting on any counter
# (s_waitcnt), SQ_WAIT_INST_ANY = waiting for an instruction to be issued / fetched, SQ_WAIT_INST_LDS = waiting for an LDS instruction to
# be issued; SQ_ACTIVE_INST_* = cycles an instruction of that kind was executing; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = share of LDS
# cycles lost to bank conflicts.
set -o pipefail
WL=${1:-full_4a}; N=${2:-10000}
O=gpurun_out/${SQ_OUT:-r5}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
: > $O/sq_stalls_$WL.txt
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  rm -rf $O/sqs_$WL
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/sqs_$WL -- python3 scripts/kernel_times.py --child $WL $N > $O/sqs_$WL.log 2>&1 || { tail -5 $O/sqs_$WL.log; exit 1; }
  echo "# $SET" >> $O/sq_stalls_$WL.txt
  python3 scripts/sq_summary.py "$O/sqs_$WL" >> $O/sq_stalls_$WL.txt
  rm -rf $O/sqs_$WL
done
cat $O/sq_stalls_$WL.txt
