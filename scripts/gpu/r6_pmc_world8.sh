#!/bin/bash
# Round 6: PMC passes of the bench command on the final build (summary + traffic JSON stamped with the build digest), then the driver's
# 8-rank command rehearsed on one GPU (gloo, labelled REHEARSAL) with the stdout check, and the N-rank runners at world 8.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_pmc; rm -rf $O; mkdir -p $O
export PGMI_GIT_HEAD=${PGMI_GIT_HEAD:-unknown}
bash scripts/pmc_profile.sh r6 --steps 1 --warmup 0 --cpu-seconds 0 --layers 4 --no-box-state --no-live-traffic > $O/pmc.log 2>&1; cp gpurun_out/pmc_r6/summary.txt $O/pmc_summary.txt; cp gpurun_out/pmc_r6/pmc_traffic.json $O/pmc_traffic.json
grep -E "^==|MfmaUtil|HBM" $O/pmc_summary.txt | head -40
find gpurun_out/pmc_r6 -name "*.csv" -delete; find gpurun_out/pmc_r6 -name "*.db" -delete
bash scripts/gpu/r6_world8.sh
