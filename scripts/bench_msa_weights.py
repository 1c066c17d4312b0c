"""Throughput of the alignment pair-count kernel (pgmi_msa_cluster_counts) on one MI355X, with the C
oracle (OpenMP, the numba kernel's restatement) timed beside it on a bounded sample.

    python scripts/bench_msa_weights.py [--n 100000] [--l 400]

Unit of work: one (i, j, column) symbol compare of the reference's ordered-pair loop (N*N*L).  Bound:
vector ALU -- 8 VALU ops per 32 columns per unordered pair (5 xor, 2 or3, 1 bcnt); the kernel visits the
upper triangle of tile pairs only (the pair distance is symmetric), so valu_frac is executed ops / peak.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteingym_amd import weights as pw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--l", type=int, default=400)
    ap.add_argument("--cpu-rows", type=int, default=0, help="sequences in the CPU sample (0 = pick for ~10 s)")
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    centers = rng.integers(0, 20, size=(a.n // 50 + 1, a.l))
    m = centers[rng.integers(0, len(centers), size=a.n)].copy()
    mask = rng.random((a.n, a.l)) < (rng.random(a.n) * 0.5)[:, None]
    m[mask] = rng.integers(0, 20, size=int(mask.sum()))
    m[rng.random((a.n, a.l)) < 0.1] = 20
    m = m.astype(np.int8)
    pw.num_cluster_members(m[:1024], 0.8, 20)                       # warm-up (module load)
    t0 = time.perf_counter()
    counts, kms = pw.num_cluster_members(m, 0.8, 20, return_ms=True)
    wall = time.perf_counter() - t0
    compares = float(a.n) * a.n * a.l
    wpad = (a.l + 31) // 32 * 32
    npad = (a.n + 127) // 128 * 128
    tiles = npad // 128
    valu_ops = tiles * (tiles + 1) / 2 * 128 * 128 * wpad / 32 * 8      # upper triangle of tile pairs only
    peak_ops = 256 * 64 * a.clock_ghz * 1e9
    out = {"n": a.n, "l": a.l, "kernel_ms": kms, "wall_s_incl_pcie_and_encode": wall,
           "compares_per_s": compares / (kms * 1e-3), "valu_ops_per_s": valu_ops / (kms * 1e-3),
           "valu_peak_ops_per_s": peak_ops, "valu_frac": valu_ops / (kms * 1e-3) / peak_ops,
           "mean_cluster": float(counts.mean())}
    try:
        from oracle import msa_weights_oracle as mo
        from bench import usable_cores
        cores = usable_cores()
        # bounded sample: a k x k sub-alignment sized for ~10 s at ~1.5e9 compares/s/core
        k = a.cpu_rows or int(min(a.n, max(2000, (10 * 1.5e9 * cores / a.l) ** 0.5)))
        sub = np.ascontiguousarray(m[:k])
        t0 = time.perf_counter()
        c_cpu = mo.cluster_counts(sub, 0.8, 20, threads=cores)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"kind": "port", "cores": cores, "sample": f"first {k} sequences (k^2*L compares)",
                               "compares_per_s": float(k) * k * a.l / dt, "seconds": dt}
        assert np.array_equal(c_cpu, pw.num_cluster_members(sub, 0.8, 20))
        out["gpu_over_cpu"] = out["compares_per_s"] / out["cpu_baseline"]["compares_per_s"]
    except ImportError as e:                                           # oracle is test infrastructure: optional here
        out["cpu_baseline"] = f"unavailable: {e}"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
