#!/bin/bash
# Everything profiles/r2/ holds, in one gpurun call: bench line + kernel trace + PMC traffic for config 4a, the bench line of
# the mix, the phase profiles and the SQ instruction counters.  Outputs under gpurun_out/r2/ (copy the summaries to profiles/r2/).
set -o pipefail
O=gpurun_out/r2; mkdir -p $O
bash scripts/profile_r2.sh full_4a > $O/profile_full_4a.log 2>&1 || { tail -5 $O/profile_full_4a.log; exit 1; }
timeout -k 10 400 python3 bench.py --workload mix --no-other-workloads > $O/bench_mix.json 2> $O/bench_mix.err || { tail -5 $O/bench_mix.err; exit 1; }
timeout -k 10 300 python3 scripts/phase_profile.py full_4a 10000 prepass > $O/phase_profile_4a.txt 2>&1 &&
timeout -k 10 300 python3 scripts/phase_profile.py mix 12500 prepass > $O/phase_profile_mix.txt 2>&1 &&
timeout -k 10 300 python3 scripts/phase_profile.py huf_literals 10000 > $O/phase_profile_huf.txt 2>&1 &&
bash scripts/sq_counters.sh full_4a 10000 > /dev/null && bash scripts/mix_trace.sh > $O/mix_kernel_timeline.txt 2>&1
tail -c 400 $O/bench_full_4a.json; echo; tail -c 300 $O/bench_mix.json; echo; cat $O/full_4a_kernel_stats.csv | head -5
