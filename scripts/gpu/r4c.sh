#!/bin/bash
# intra-XCD start stagger: CU slot k starts k x s x 0.1 us late (variants 1512 + 64 s)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c; mkdir -p $O
timeout 600 python scripts/gemm_ab.py 5 1512 1576 1640 1704 1768 1960 > $O/gemm_stagger_cu.log 2>&1
cat $O/gemm_stagger_cu.log
