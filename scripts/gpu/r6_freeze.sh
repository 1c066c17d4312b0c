#!/bin/bash
# Round 6: run the -m gpu suite once with PGMI_FREEZE_DIR set: the CPU oracle's outputs at the real widths are written as npz (tests/frozen.py),
# copied to tests/golden/frozen/ afterwards; later runs read them instead of re-running the oracle.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_freeze; rm -rf $O gpurun_out/frozen; mkdir -p $O
PGMI_FREEZE_DIR=gpurun_out/frozen timeout 2400 python -m pytest tests -q -m gpu --durations=25 > $O/gpu_suite_freeze.log 2>&1; echo "rc $?" >> $O/gpu_suite_freeze.log; tail -40 $O/gpu_suite_freeze.log
ls -la gpurun_out/frozen; du -sh gpurun_out/frozen
