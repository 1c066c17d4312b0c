#!/bin/bash
# Round 6, call 2: K = 15 tiles (ESM2-35M's width) in f16x3; the two-role attention kernel: bits, then the interleaved A/B by shape.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_call2; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm_f16x3" > $O/gemm_k480.log 2>&1; echo "rc $?" >> $O/gemm_k480.log; tail -5 $O/gemm_k480.log
timeout 600 python -m pytest tests/test_gpu_esm.py -q -x -s -k "35m_width or small_head or score or all_positions" > $O/esm2_35m.log 2>&1; echo "rc $?" >> $O/esm2_35m.log; tail -8 $O/esm2_35m.log
bash scripts/gpu/r6_att.sh call2_att
timeout 300 python -m pytest tests/test_gpu_esm.py -q -x -k "attention_launch_options" > $O/att_model.log 2>&1; echo "rc $?" >> $O/att_model.log; tail -5 $O/att_model.log
