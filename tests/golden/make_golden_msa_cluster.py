"""Golden vectors for the sequence-weight pair count, produced by the REFERENCE's own functions
(proteingym/utils/weights.py: calc_num_cluster_members_nogaps, ..._parallel, calc_weights_fast) run in
pure-python mode through oracle/ref_harness.load_reference_weights (numba stubbed to identity).

    python tests/golden/make_golden_msa_cluster.py        ->  tests/golden/golden_msa_cluster.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_harness as rh  # noqa: E402


def make_case(rng, n, l, gap_rate, n_clusters, mut_rate, empty_rows=0, gap=20):
    centers = rng.integers(0, 20, size=(n_clusters, l))
    m = centers[rng.integers(0, n_clusters, size=n)].copy()
    mut = rng.random((n, l)) < mut_rate[:, None] if isinstance(mut_rate, np.ndarray) else rng.random((n, l)) < mut_rate
    m[mut] = rng.integers(0, 20, size=int(mut.sum()))
    m[rng.random((n, l)) < gap_rate] = gap
    for r in rng.choice(n, size=empty_rows, replace=False):
        m[r] = gap
    return m.astype(np.int64)


def main():
    w = rh.load_reference_weights()
    rng = np.random.default_rng(20240917)
    cases = {
        "small": (make_case(rng, 40, 17, 0.1, 3, 0.15), 0.8),
        "ragged": (make_case(rng, 130, 45, 0.25, 5, rng.random(130) * 0.4, empty_rows=3), 0.8),
        "wide": (make_case(rng, 70, 100, 0.05, 4, 0.08), 0.8),
        "thr_edge": (make_case(rng, 90, 10, 0.2, 2, 0.2), 0.8),          # matches/nongap hits 0.8 exactly: strict '>'
        "thr_07": (make_case(rng, 60, 33, 0.1, 3, 0.3), 0.7),
        "thr_1m02": (make_case(rng, 60, 20, 0.3, 3, 0.2), 1 - 0.2),      # the caller's spelling of the threshold
    }
    out = {}
    for name, (m, thr) in cases.items():
        empty = w.is_empty_sequence_matrix(m, empty_value=20)
        serial = w.calc_num_cluster_members_nogaps(m[~empty], thr, 20)
        par = w.calc_num_cluster_members_nogaps_parallel(m[~empty], thr, 20)
        assert np.array_equal(serial, par)
        counts = np.zeros(len(m))
        counts[~empty] = par
        out[f"{name}/matrix"] = m.astype(np.int8)
        out[f"{name}/threshold"] = np.float64(thr)
        out[f"{name}/counts"] = counts.astype(np.int32)
        out[f"{name}/weights"] = w.calc_weights_fast(m, thr, 20, num_cpus=1)
        print(name, m.shape, "empty", int(empty.sum()), "max cluster", int(counts.max()))
    np.savez_compressed(os.path.join(HERE, "golden_msa_cluster.npz"), **out)


if __name__ == "__main__":
    main()
