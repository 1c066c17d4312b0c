#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_msa_transformer.py -x -q -m gpu > $O/test_msa.log 2>&1; tail -5 $O/test_msa.log
timeout 600 python scripts/bench_msa_transformer.py > $O/bench_msa_287.json 2> $O/bench_msa_287.err; python -c "
import json; d=json.loads(open('$O/bench_msa_287.json').read().strip().splitlines()[-1]); print('287:', d['ms_per_forward'], d['tflops_algorithmic'], {k:v['ms_per_forward'] for k,v in d['profile'].items()})"
timeout 600 python scripts/bench_msa_transformer.py --cols 1024 --positions 3 > $O/bench_msa_1024.json 2> $O/bench_msa_1024.err; python -c "
import json; d=json.loads(open('$O/bench_msa_1024.json').read().strip().splitlines()[-1]); print('1024:', d['ms_per_forward'], d['tflops_algorithmic'], {k:v['ms_per_forward'] for k,v in d['profile'].items()})"
timeout 300 python scripts/gemm_ab.py 3 0 > $O/gemm_ab.log 2>&1; cat $O/gemm_ab.log
