#!/bin/bash
# Round 5, first GPU call: the prefix-shared Tranception path (bit-identity tests), everything the attention / option changes touch,
# the Tranception bench (every sequence in full vs prefix-shared).        bash scripts/gpu/r5_call1.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r5_call1; rm -rf $O; mkdir -p $O
(rocm-smi --showpower --showclocks) > $O/box.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_tranception.py -q -m gpu -x > $O/tests_tranception.log 2>&1; echo "tranception rc $?"; tail -5 $O/tests_tranception.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_esm.py tests/test_gpu_real_shape.py tests/test_gpu_pppl.py -q -m gpu > $O/tests_ops_esm.log 2>&1; echo "ops/esm rc $?"; tail -5 $O/tests_ops_esm.log
timeout 400 python scripts/bench_tranception.py > $O/bench_tranception.json 2> $O/bench_tranception.err; echo "bench rc $?"; cat $O/bench_tranception.json | cut -c1-1500
