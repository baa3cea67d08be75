#!/bin/bash
# phase shares of the diagnostic build (make prof): scripts/pp.sh <out-name> <workload> <frames>
mkdir -p gpurun_out/r4
timeout -k 10 300 python scripts/phase_profile.py $2 $3 prepass > gpurun_out/r4/$1.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r4/$1.log | tail -40
exit $rc
