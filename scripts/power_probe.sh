#!/bin/bash
# sample power / clocks while a GEMM loop runs (GPU box)
prec=${1:-f16x3}; var=${2:-2}
python - <<PY &
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from proteingym_amd import _lib
lib = _lib.load()
ms = C.c_double()
for rep in range(8):
    _lib.check(lib.pgmi_bench_gemm(0, _lib.PRECISIONS["$prec"], 82368, 5120, 1280, 1, 0 if "$prec"=="fp32" else 1, $var, 400, C.byref(ms)))
    print("fc1", "$prec", $var, 2.0*82368*5120*1280/ms.value/1e9, "TF", flush=True)
PY
sleep 3
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | head -6; echo ---; sleep 1; done
wait
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
