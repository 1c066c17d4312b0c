#!/bin/bash
# Round 5: the loop-structure probe (tools/mfma_duo.hip), interleaved rounds on one box, clock / power sampled beside it.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5_probe; rm -rf $O; mkdir -p $O
(while true; do rocm-smi --showpower --showclocks | grep -E "sclk|Socket Power|Power \(W\)"; sleep 1; done) > $O/smi.log 2>&1 &
SMI=$!
timeout 300 tools/mfma_duo 6 > $O/mfma_duo.log 2>&1; echo "probe rc $?"
kill $SMI
cat $O/mfma_duo.log | tail -12
grep -E "sclk" $O/smi.log | awk '{print $NF}' | sort | uniq -c | sort -rn | head -5
