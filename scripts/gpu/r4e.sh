#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_outliers.py -x -q -m gpu -s > $O/test_outliers.log 2>&1; echo "rc $?" >> $O/test_outliers.log
grep -E "outlier checkpoint|overflow case|passed|failed|rc |Error|assert" $O/test_outliers.log | head -30
timeout 1500 python -m pytest tests/test_gpu_parity_real_width.py -q -m gpu -s -k "pppl_735" > $O/test_pppl.log 2>&1; echo "rc $?" >> $O/test_pppl.log
grep -E "pseudo-ppl|passed|failed|rc |Error|assert" $O/test_pppl.log | head -30
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_esm.py -x -q -m gpu > $O/test_ops_esm.log 2>&1; tail -3 $O/test_ops_esm.log
