#!/bin/bash
# SQ instruction counters of the real-data diagnostic (one rocprofv3 --pmc pass, no trace domains)
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf $O/sq_real
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_real -- python3 scripts/real_data_bench.py 12000 3 > $O/sq_real.log 2>&1 || { tail -5 $O/sq_real.log; exit 1; }
python3 scripts/sq_summary.py "$O/sq_real" > $O/sq_real.txt
rm -rf $O/sq_real
grep "GPU step" $O/sq_real.log; cat $O/sq_real.txt
