#!/bin/bash
# Round 6, call 1: the new failure-isolation GPU test, the attention baseline of this round's box, the whole -m gpu suite with durations.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_call1; rm -rf $O; mkdir -p $O
(rocm-smi --showpower --showclocks) > $O/box.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_outliers.py -q -x -s -k "run_benchmark" > $O/outlier_retry.log 2>&1; echo "rc $?" >> $O/outlier_retry.log; tail -5 $O/outlier_retry.log
timeout 300 python scripts/att_bench.py --rounds 5 > $O/att_bench_baseline.log 2>&1; tail -6 $O/att_bench_baseline.log
timeout 1500 python -m pytest tests -q -m gpu --durations=40 > $O/gpu_suite.log 2>&1; echo "rc $?" >> $O/gpu_suite.log; tail -60 $O/gpu_suite.log
