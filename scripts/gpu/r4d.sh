#!/bin/bash
# attention: one workgroup per (sequence, head) with 8 / 9 waves (every query tile of T = 288 in one block: K / V^T read once)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4d; mkdir -p $O
for w in 9 8; do PGMI_ATT_WPB=$w timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" > $O/test_att_wpb$w.log 2>&1; tail -2 $O/test_att_wpb$w.log; done
timeout 600 python scripts/att_bench.py --rounds 5 --configs 0:4:0:0,0:4:9:0,0:4:8:0 > $O/att_bench.log 2>&1
grep -v "^{" $O/att_bench.log
