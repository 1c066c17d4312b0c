#!/bin/bash
# Round 6: the -m gpu suite as the driver runs it (oracle outputs at the real widths from tests/golden/frozen), with durations.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_suite; rm -rf $O; mkdir -p $O
(rocm-smi --showpower --showclocks) > $O/box.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu --durations=15 > $O/gpu_suite.log 2>&1; echo "rc $?" >> $O/gpu_suite.log; tail -30 $O/gpu_suite.log
