#!/bin/bash
# Everything profiles/r3/ holds, in one gpurun call: for config 4a (headline), config 3 (huf_literals) and the mix the bench line,
# the rocprofv3 kernel-trace summary of the same command and the two PMC passes (FETCH_SIZE, WRITE_SIZE: runs of their own); the
# bench line of config 2; the phase profile of the mix and the SQ instruction counters of 4a and config 3.
# Outputs under gpurun_out/r3/ (the summaries are then copied to profiles/r3/).
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
for WL in full_4a huf_literals mix; do
  EXTRA=""; [ $WL != full_4a ] && EXTRA="--no-other-workloads"
  echo "== $WL"
  cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
  timeout -k 10 500 python3 bench.py --workload $WL $EXTRA > $O/bench_$WL.json 2> $O/bench_$WL.err || { tail -5 $O/bench_$WL.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$WL -- python3 bench.py --workload $WL --no-cpu-baseline --no-other-workloads > $O/bench_${WL}_under_rocprof.json 2> $O/kt_$WL.err || { tail -5 $O/kt_$WL.err; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$WL -- python3 bench.py --workload $WL --no-cpu-baseline --no-other-workloads --no-verify-all --steps 3 --warmup 1 > $O/pmc_fetch_$WL.json 2> $O/pmc_fetch_$WL.err || { tail -5 $O/pmc_fetch_$WL.err; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$WL -- python3 bench.py --workload $WL --no-cpu-baseline --no-other-workloads --no-verify-all --steps 3 --warmup 1 > $O/pmc_write_$WL.json 2> $O/pmc_write_$WL.err || { tail -5 $O/pmc_write_$WL.err; exit 1; }
  python3 scripts/pmc_summary.py $O/pmc_fetch_$WL $O/pmc_write_$WL $O/bench_${WL}_under_rocprof.json $O/pmc_hbm_traffic_$WL.json > /dev/null
  find $O/kt_$WL -name "*kernel_stats.csv" -exec cp {} $O/${WL}_kernel_stats.csv \;
  rm -rf $O/pmc_fetch_$WL $O/pmc_write_$WL $O/kt_$WL
  tail -c 300 $O/bench_$WL.json; echo
done
timeout -k 10 300 python3 bench.py --workload raw_rle --no-other-workloads > $O/bench_raw_rle.json 2> $O/bench_raw_rle.err
timeout -k 10 300 python3 scripts/phase_profile.py mix 12500 prepass > $O/phase_profile_mix.txt 2>&1
timeout -k 10 300 python3 scripts/phase_profile.py full_4a 10000 prepass > $O/phase_profile_4a.txt 2>&1
bash scripts/sq_counters.sh full_4a 10000 > /dev/null; bash scripts/sq_counters.sh huf_literals 10000 > /dev/null
head -8 $O/full_4a_kernel_stats.csv
