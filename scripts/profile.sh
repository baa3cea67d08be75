#!/bin/bash
# Everything profiles/<round>/ holds for a workload, in one gpurun call:  scripts/profile.sh <round> [workload ...]
#   bench line, rocprofv3 --kernel-trace --stats summary of the same command, the two PMC passes (FETCH_SIZE, WRITE_SIZE: runs of
#   their own, never combined with a trace domain) and the SQ instruction counters -> gpurun_out/<round>/ (copy the summaries to
#   profiles/<round>/).  Default workloads: full_4a huf_literals mix.  PMC_EXTRA: extra bench.py flags for the PMC passes, with the
#   outputs suffixed _alt (e.g. PMC_EXTRA=--no-wexec-kernel: under a counter pass the kernels of a step run one after the other, so
#   cz_wexec_kernel, first in line, takes every frame it is listed; the alt pass shows cz_execute_frames_kernel's traffic).
set -o pipefail
R=${1:-r5}; shift
WLS=${@:-full_4a huf_literals mix}
O=gpurun_out/$R; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for WL in $WLS; do
  EXTRA=""; [ $WL != full_4a ] && EXTRA="--no-other-workloads"
  echo "== $WL"
  timeout -k 10 600 python3 bench.py --workload $WL $EXTRA > $O/bench_$WL.json 2> $O/bench_$WL.err || { tail -5 $O/bench_$WL.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$WL -- python3 bench.py --workload $WL --no-cpu-baseline --no-other-workloads > $O/bench_${WL}_under_rocprof.json 2> $O/kt_$WL.err || { tail -5 $O/kt_$WL.err; exit 1; }
  find $O/kt_$WL -name "*kernel_stats.csv" -exec cp {} $O/${WL}_kernel_stats.csv \;
  for ALT in "" $PMC_EXTRA; do
    SFX=""; [ -n "$ALT" ] && SFX="_alt"
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$WL -- python3 bench.py --workload $WL --no-cpu-baseline --no-other-workloads --no-verify-all --steps 3 --warmup 1 $ALT > $O/pmc_fetch_$WL$SFX.json 2> $O/pmc_fetch_$WL.err || { tail -5 $O/pmc_fetch_$WL.err; exit 1; }
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$WL -- python3 bench.py --workload $WL --no-cpu-baseline --no-other-workloads --no-verify-all --steps 3 --warmup 1 $ALT > $O/pmc_write_$WL$SFX.json 2> $O/pmc_write_$WL.err || { tail -5 $O/pmc_write_$WL.err; exit 1; }
    python3 scripts/pmc_summary.py $O/pmc_fetch_$WL $O/pmc_write_$WL $O/pmc_fetch_$WL$SFX.json $O/pmc_hbm_traffic_$WL$SFX.json > /dev/null
    rm -rf $O/pmc_fetch_$WL $O/pmc_write_$WL
  done
  rm -rf $O/kt_$WL
  SQ_OUT=$R bash scripts/sq_counters.sh $WL $( [ $WL = mix ] && echo 12500 || echo 10000 ) > /dev/null
  tail -c 300 $O/bench_$WL.json; echo
done
head -12 $O/full_4a_kernel_stats.csv 2>/dev/null
