"""What the FC1-class epilogue costs: the same GEMM (M x 5120 x 1280, and the FC2 shape for reference) with each output kind.
    python scripts/gemm_epilogue_cost.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from proteingym_amd import _lib

lib = _lib.load()
M = int(os.environ.get("GEMM_M", 82368))
v = np.array([0], dtype=np.int32)
for name, N, K in (("fc1 shape", 5120, 1280), ("fc2 shape", 1280, 5120)):
    for epi, split, what in ((0, 0, "fp32 out"), (1, 0, "GELU, fp32 out"), (0, 1, "split-plane out"), (1, 1, "GELU, split-plane out")):
        out = np.zeros(1, dtype=np.float64)
        _lib.check(lib.pgmi_bench_gemm_ab(0, _lib.PRECISIONS["f16x3"], M, N, K, epi, split, _lib.ptr(v, _lib._i32p), 1, 3,
                                          int(os.environ.get("GEMM_ITERS", 25)), _lib.ptr(out, _lib._f64p)))
        print(f"{name}  {what:24s} {out[0]:.3f} ms  {2.0 * M * N * K / out[0] / 1e9:6.1f} TFLOP/s", flush=True)
