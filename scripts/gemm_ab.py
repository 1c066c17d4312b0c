"""Within-process interleaved A/B of GEMM variants (perf deltas come from interleaved rounds in ONE process on ONE set of
operands).   python scripts/gemm_ab.py ROUNDS v1 v2 ...   -> per shape median TFLOP/s per variant + layer-sum"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from proteingym_amd import _lib

lib = _lib.load()
rounds = int(sys.argv[1])
variants = np.array([int(v) for v in sys.argv[2:]], dtype=np.int32)
M = int(os.environ.get("GEMM_M", 82368))
D = int(os.environ.get("GEMM_D", 1280))
# output kinds as in the product: fused QKV epilogue, in-place residual, GELU + split planes
shapes = [("qkv", 3 * D, D, 0, 3), ("out", D, D, 0, 2), ("fc1", 4 * D, D, 1, 1), ("fc2", D, 4 * D, 0, 2)]
ms = {}
for name, N, K, epi, split in shapes:
    out = np.zeros(len(variants), dtype=np.float64)
    _lib.check(lib.pgmi_bench_gemm_ab(0, _lib.PRECISIONS["f16x3"], M, N, K, epi, split, _lib.ptr(variants, _lib._i32p),
                                      len(variants), rounds, int(os.environ.get("GEMM_ITERS", 25)), _lib.ptr(out, _lib._f64p)))
    ms[name] = out
for k, v in enumerate(variants):
    tot_ms = sum(ms[n][k] for n, *_ in shapes)
    tot_fl = sum(2.0 * M * N * K for _, N, K, _, _ in shapes)
    row = "  ".join(f"{n} {2.0 * M * N * K / ms[n][k] / 1e9:6.1f}" for n, N, K, _, _ in shapes)
    print(f"variant {v:4d}: {row}   layer-sum {tot_fl / tot_ms / 1e9:6.1f} TF ({tot_ms:.2f} ms)", flush=True)
