#!/bin/bash
# Round 6: the new tests (run_indels' fp32 redo, the early range check of the pseudo-ppl call, the seeded shape sweep), then a wider
# sweep with another seed for discovery.       bash scripts/gpu/r6_sweep.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_sweep; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_outliers.py -m gpu -q -x -s -k "run_indels or run_benchmark" > $O/outliers.log 2>&1; echo "outliers rc $?"; tail -5 $O/outliers.log
timeout 900 python -m pytest tests/test_gpu_random_shapes.py -m gpu -q -s --durations=5 > $O/sweep_default.log 2>&1; echo "sweep default rc $?"; tail -12 $O/sweep_default.log
PGMI_SWEEP_CASES=${WIDE:-300} PGMI_SWEEP_SEED=${SWEEP_SEED:-7} timeout 2400 python -m pytest tests/test_gpu_random_shapes.py -m gpu -q -s > $O/sweep_wide.log 2>&1; echo "sweep wide rc $?"; tail -5 $O/sweep_wide.log
grep -E "FAILED|Error" $O/sweep_wide.log | head -40
