#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_msa_transformer.py -x -q -m gpu -s -k "real_shape" > $O/test_msa_real.log 2>&1; grep -E "MSA Transformer 12|passed|failed|Error|assert" $O/test_msa_real.log | head
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_esm.py tests/test_gpu_tranception.py tests/test_gpu_msa_transformer.py -x -q -m gpu > $O/test_quick.log 2>&1; tail -3 $O/test_quick.log
timeout 600 python scripts/att_bench.py --rounds 5 --configs 0 > $O/att_bench.log 2>&1
grep -v "^{" $O/att_bench.log
