"""Shader clock and socket power next to time, per GEMM launch-parameter variant: every leg is one variant running one shape
back to back for a few seconds while `rocm-smi --showpower --showclocks` is sampled beside it.

    python scripts/gemm_power_legs.py SHAPE v1 v2 ...      SHAPE: qkv | out | fc1 | fc2   (BLAT shape, ESM-1v 650M layer)
"""
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from proteingym_amd import _lib

lib = _lib.load()
M = int(os.environ.get("GEMM_M", 82368))
D = int(os.environ.get("GEMM_D", 1280))
SHAPES = {"qkv": (3 * D, D, 0, 3), "out": (D, D, 0, 2), "fc1": (4 * D, D, 1, 1), "fc2": (D, 4 * D, 0, 2)}
shape = sys.argv[1]
N, K, epi, split = SHAPES[shape]
iters = int(os.environ.get("GEMM_ITERS", 1500 if shape != "out" else 5000))


def sample():
    try:
        txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:  # noqa: BLE001
        return None, None, repr(e)
    sclk = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)\s*Mhz", txt, re.I)
    pw = re.search(r"Power \(W\):\s*([\d.]+)", txt)
    return (int(sclk.group(1)) if sclk else None), (float(pw.group(1)) if pw else None), txt


for v in [int(x) for x in sys.argv[2:]]:
    out = np.zeros(1, dtype=np.float64)
    var = np.array([v], dtype=np.int32)
    done = threading.Event()

    def leg():
        _lib.check(lib.pgmi_bench_gemm_ab(0, _lib.PRECISIONS["f16x3"], M, N, K, epi, split, _lib.ptr(var, _lib._i32p), 1, 1, iters,
                                          _lib.ptr(out, _lib._f64p)))
        done.set()

    th = threading.Thread(target=leg)
    t0 = time.time()
    th.start()
    clocks, watts, raw = [], [], None
    while not done.wait(0.4):
        c, w, txt = sample()
        if time.time() - t0 > 3.0:          # operands are generated and uploaded in the first seconds
            if c:
                clocks.append(c)
            if w:
                watts.append(w)
            raw = txt
    th.join()
    tf = 2.0 * M * N * K / out[0] / 1e9
    print(f"{shape} variant {v:4d}: {out[0]:.4f} ms  {tf:6.1f} TFLOP/s   sclk {np.mean(clocks) if clocks else float('nan'):.0f} MHz "
          f"(n={len(clocks)})  power {np.mean(watts) if watts else float('nan'):.0f} W", flush=True)
    if not clocks and raw:
        print("rocm-smi output not parsed:\n" + raw[:1500], flush=True)
