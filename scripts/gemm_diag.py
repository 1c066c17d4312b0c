"""Phase timing of the persistent ping-pong GEMM (tuning-only instantiation, variant 13): prints, for the early and late
wave of SIMD 0 of workgroup 0, the mean shader clocks spent working in / waiting at the end of each of the four phases."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
from proteingym_amd import _lib

lib = _lib.load()
M = int(os.environ.get("GEMM_M", 82368))
for name, N, K in (("fc2", 1280, 5120), ("out", 1280, 1280)):
    ms = C.c_double()
    print(name, flush=True)
    _lib.check(lib.pgmi_bench_gemm(0, _lib.PRECISIONS["f16x3"], M, N, K, 0, 0, 13, 2, C.byref(ms)))
    print(f"  {2.0 * M * N * K / ms.value / 1e9:.1f} TF under instrumentation", flush=True)
    break
