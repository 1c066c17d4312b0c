#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r5_att; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_esm.py -q -m gpu -x -k "launch_options" > $O/tests.log 2>&1; echo "tests rc $?"; tail -12 $O/tests.log
timeout 600 python scripts/att_bench.py --rounds 7 --ab att_persist=0,att_persist=1,att_persist=2,att_persist=4 > $O/att_bench_ab.log 2>&1; echo "att_bench rc $?"; grep -v "^{" $O/att_bench_ab.log | tail -18
