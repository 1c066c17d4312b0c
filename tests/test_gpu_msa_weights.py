"""GPU parity of the alignment pair-count kernel (csrc/msa_weights.hip, through the C ABI) against the
reference-generated golden counts and the C oracle: bit-exact (integer work)."""
import os

import numpy as np
import pytest

from oracle import msa_weights_oracle as mo
from proteingym_amd import weights as pw, tranception as ptr, _lib

pytestmark = pytest.mark.gpu
CASES = ["small", "ragged", "wide", "thr_edge", "thr_07", "thr_1m02"]


def _clustered(rng, n, l, n_clusters, mut, gap_rate, gap=20):
    centers = rng.integers(0, 20, size=(n_clusters, l))
    m = centers[rng.integers(0, n_clusters, size=n)].copy()
    rate = rng.random(n) * mut
    mask = rng.random((n, l)) < rate[:, None]
    m[mask] = rng.integers(0, 20, size=int(mask.sum()))
    m[rng.random((n, l)) < gap_rate] = gap
    return m.astype(np.int8)


@pytest.mark.parametrize("name", CASES)
def test_counts_match_reference_golden(lib, golden_dir, name):
    g = np.load(os.path.join(golden_dir, "golden_msa_cluster.npz"))
    m, thr = g[f"{name}/matrix"], float(g[f"{name}/threshold"])
    assert np.array_equal(pw.num_cluster_members(m, thr, 20), g[f"{name}/counts"])
    assert np.array_equal(pw.calc_weights_fast(m, thr, 20, num_cpus=4), g[f"{name}/weights"])


@pytest.mark.parametrize("n,l", [(1, 1), (2, 31), (127, 32), (128, 33), (129, 64), (300, 65), (1000, 257), (2500, 96), (4097, 40)])
def test_counts_match_oracle_ragged_sizes(lib, n, l):
    rng = np.random.default_rng(n * 1000 + l)
    m = _clustered(rng, n, l, max(1, n // 40), 0.5, 0.15)
    if n > 10:
        m[rng.integers(0, n, size=3)] = 20                       # empty rows
    for thr in (0.8, 0.5):
        assert np.array_equal(pw.num_cluster_members(m, thr, 20), mo.cluster_counts(m, thr, 20))


def test_other_invalid_value_and_symbols_up_to_29(lib):
    rng = np.random.default_rng(5)
    m = rng.integers(0, 30, size=(400, 50)).astype(np.int8)
    m[rng.random(m.shape) < 0.2] = -1
    m[100:140] = m[7]
    assert np.array_equal(pw.num_cluster_members(m, 0.8, -1), mo.cluster_counts(m, 0.8, -1))


def test_errors(lib):
    m = np.zeros((4, 4), dtype=np.int8)
    m[0, 0] = 31
    with pytest.raises(_lib.PgmiError, match="outside 0..29"):
        pw.num_cluster_members(m, 0.8, 20)
    with pytest.raises(_lib.PgmiError, match="identity_threshold"):
        pw.num_cluster_members(np.zeros((4, 4), dtype=np.int8), 1.5, 20)


def test_large_alignment_properties(lib):
    """Size-independent checks at a size the scalar oracle would need minutes for: a row permutation
    permutes the counts; appending exact copies of k rows adds k-per-copy to their clusters only; a
    sampled subset of rows agrees with the C oracle evaluated for those rows."""
    rng = np.random.default_rng(11)
    n, l = 30000, 250
    m = _clustered(rng, n, l, 300, 0.5, 0.1)
    c = pw.num_cluster_members(m, 0.8, 20)
    perm = rng.permutation(n)
    assert np.array_equal(pw.num_cluster_members(m[perm], 0.8, 20), c[perm])
    assert c.min() >= 1 and (c > 1).any()
    rows = rng.choice(n, size=64, replace=False)
    valid = m != 20
    for r in rows[:16]:
        matches = ((m == m[r]) & valid[r]).sum(1)
        keep = matches / valid[r].sum() > 0.8
        keep[r] = True
        assert int(keep.sum()) == int(c[r])
    dup = np.concatenate([m, m[rows]])
    c2 = pw.num_cluster_members(dup, 0.8, 20)
    assert np.array_equal(c2[n:], c2[rows])
    assert (c2[rows] >= c[rows] + 1).all()


def test_msa_processing_on_device_matches_reference_weights(lib, golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "golden_msa_weights.npz"))
    mp = ptr.MSA_processing(MSA_location=os.path.join(golden_dir, "TOY_MSA_GAPPY.a2m"), use_weights=True,
                            weights_location=str(tmp_path / "w.npy"), device=0)
    assert np.array_equal(mp.weights, g["weights"])
    assert list(mp.seq_name_to_weight.keys()) == list(g["names"]) and abs(mp.Neff - float(g["Neff"])) < 1e-12
    assert np.array_equal(np.load(str(tmp_path / "w.npy")), g["weights"])          # computed once, saved where the reference saves it
    again = ptr.MSA_processing(MSA_location=os.path.join(golden_dir, "TOY_MSA_GAPPY.a2m"), use_weights=True,
                               weights_location=str(tmp_path / "w.npy"))             # default device; the file is loaded
    assert np.array_equal(again.weights, g["weights"])
