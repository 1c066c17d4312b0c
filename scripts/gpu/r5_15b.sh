#!/bin/bash
# Round 5: ESM2-15B instantiated once at full depth (48 x 5120 x 40 heads of 128: 60 GB of weight planes) -- tests/test_gpu_parity_real_width.py
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r5_15b; rm -rf $O; mkdir -p $O
PGMI_TEST_15B_FULL=1 timeout 1500 python -m pytest tests/test_gpu_parity_real_width.py -q -m gpu -s -k "15b_full_depth" > $O/test_15b_full_depth.log 2>&1; echo "rc $?"
grep -E "ESM2-15B|passed|failed|Error|error" $O/test_15b_full_depth.log | tail -8
rocm-smi --showmeminfo vram | grep -i "used" | head -2
