"""GPU tuning loop for the GEMM kernels: TFLOP/s per (precision, variant, shape).
    python scripts/gemm_tune.py [precision] [variants...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
from proteingym_amd import _lib

lib = _lib.load()
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
variants = [int(v) for v in sys.argv[2:]] or [0]
M = 82368
shapes = [("qkv", 3840, 1280, 0, 0), ("out", 1280, 1280, 0, 0), ("fc1", 5120, 1280, 1, 1), ("fc2", 1280, 5120, 0, 0)]
for v in variants:
    row = []
    tot_ms, tot_fl = 0.0, 0.0
    for name, N, K, epi, split in shapes:
        ms = C.c_double()
        _lib.check(lib.pgmi_bench_gemm(0, _lib.PRECISIONS[prec], M, N, K, epi, split if prec != "fp32" else 0, v, 5, C.byref(ms)))
        fl = 2.0 * M * N * K
        row.append(f"{name} {fl / ms.value / 1e9:7.1f}")
        tot_ms += ms.value
        tot_fl += fl
    print(f"{prec} variant {v}: " + "  ".join(row) + f"   layer-sum {tot_fl / tot_ms / 1e9:7.1f} TF ({tot_ms:.2f} ms)", flush=True)
