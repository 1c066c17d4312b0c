#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r5_call4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tranception.py tests/test_gpu_two_ranks.py -q -m gpu -x -k "tranception or prefix or score_mutants or token_logprobs or retrieval" > $O/tests_tranception.log 2>&1; echo "tranception rc $?"; tail -5 $O/tests_tranception.log
timeout 400 python scripts/bench_tranception.py > $O/bench_tranception.json 2> $O/bench_tranception.err; echo "bench rc $?"; cut -c1-900 $O/bench_tranception.json
free -g | head -2; nproc
