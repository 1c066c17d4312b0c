#!/bin/bash
# Round 6: the two-role attention kernel -- bits against the 4-wave kernel, then the interleaved A/B by shape.   bash scripts/gpu/r6_att.sh [tag]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-att}
O=gpurun_out/r6_$TAG; rm -rf $O; mkdir -p $O
(rocm-smi --showpower --showclocks) > $O/box.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" > $O/att_tests.log 2>&1; echo "rc $?" >> $O/att_tests.log; tail -15 $O/att_tests.log
timeout 400 python scripts/att_bench.py --rounds 7 --shapes 286x286,90x1100,150x150,600x120,200x230,120x500,60x737 --ab att_v3=0,att_v3=1 > $O/att_ab.log 2>&1; cat $O/att_ab.log | grep -v "^{"
