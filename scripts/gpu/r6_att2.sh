#!/bin/bash
# Round 6: attention v3 epilogue variants (att_v3 = 1: lane-exchange stores, 2: through LDS) against v2, model-level bits first.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_att2; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_esm.py tests/test_gpu_ops.py -q -x -k "attention" > $O/att_tests.log 2>&1; echo "rc $?" >> $O/att_tests.log; tail -4 $O/att_tests.log
timeout 400 python scripts/att_bench.py --rounds 7 --shapes 286x286,90x1100,200x230,120x500,60x737 --ab ${1:-att_v3=0,att_v3=1,att_v3=2} > $O/att_ab.log 2>&1; grep -v "^{" $O/att_ab.log
