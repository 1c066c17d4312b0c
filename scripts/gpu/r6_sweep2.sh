#!/bin/bash
# Round 6: the runner sweep, then wide draws of the whole sweep with further seeds.       bash scripts/gpu/r6_sweep2.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_sweep2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_random_shapes.py -m gpu -q -x -k runner > $O/runner_default.log 2>&1; echo "runner default rc $?"; tail -5 $O/runner_default.log
for seed in ${SEEDS:-21 22}; do
  PGMI_SWEEP_CASES=${WIDE:-360} PGMI_SWEEP_SEED=$seed timeout 2400 python -m pytest tests/test_gpu_random_shapes.py -m gpu -q -s > $O/sweep_seed$seed.log 2>&1; echo "sweep seed $seed rc $?"; tail -3 $O/sweep_seed$seed.log
  grep -E "^FAILED" $O/sweep_seed$seed.log | head -20
done
