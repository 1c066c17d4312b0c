#!/bin/bash
# Round 6: what bounds a segment of the two-role attention kernel -- timing probes (wrong results), interleaved with the real kernel and the 4-wave one.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_att_probe; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "attention_v3" > $O/att_tests.log 2>&1; echo "rc $?" >> $O/att_tests.log; tail -3 $O/att_tests.log
timeout 600 python scripts/att_bench.py --rounds 5 --shapes 286x286,90x1100 --ab ${1:-att_v3=0,att_v3=1} > $O/att_probe.log 2>&1; grep -v "^{" $O/att_probe.log
