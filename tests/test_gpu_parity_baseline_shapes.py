"""Parity at BASELINE.json's own shapes, flat 1e-4 abs against the fp32 CPU oracle (the north-star bar:
"per-mutant scores match the reference CPU ESM-1v path within 1e-4 abs"), in both parity-gated precision modes.

  * config 2: ESM-1v 650M (33 x 1280 x 20 x 5120), a BLAT_ECOLX-shaped assay (L=286 -> T=288): every one of the
    286 table rows the assay reads and the full 4 996-single + 400 multi-mutant (depth 2-5) score vector;
  * config 3: ESM2-3B at FULL depth (36 x 2560 x 40 x 10240), the same protein length, rows of the positions two
    depth-5 mutants and a set of singles read, and their scores.

Weights: ``embed_tokens ~ N(0, 0.15^2)`` (gives |LLR| of the size real ESM checkpoints
produce); the N(0,1) default-init case is a *reported* stress number (relative bar), not the gate.
The oracle (oracle/esm_oracle.py) follows /root/reference/proteingym/baselines/esm/esm/model/esm1.py:116-177,
esm2.py:76-143 and compute_fitness.py:240-250,486-514; it is pinned to the reference by tests/test_oracle_pinning.py.
"""
import numpy as np
import pytest

from proteingym_amd import esm as pesm, synthetic

import frozen

pytestmark = pytest.mark.gpu
TOL = 1e-4
L_BLAT = 286

_CACHE = {}


def _weights(name, embed_std, seed):
    key = (name, embed_std, seed)
    if key not in _CACHE:
        _CACHE.clear()                                    # one multi-GB blob at a time
        cfg = dict(getattr(synthetic, name))
        _CACHE[key] = (cfg, synthetic.random_weights(cfg, seed=seed, embed_std=embed_std))
    return _CACHE[key]


_ORACLE = {}


def _oracle_table(name, embed_std, seed, seq, positions, batch):
    """fp32 CPU oracle rows, computed once per (model, sequence) and shared by the precision modes."""
    import torch
    from oracle import esm_oracle as eo
    key = (name, embed_std, seed, seq, tuple(positions))
    if key not in _ORACLE:
        _ORACLE.clear()
        cfg, blob = _weights(name, embed_std, seed)

        def compute():
            torch.set_num_threads(max(1, __import__("bench").usable_cores()))
            ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), **cfg)
            return {"table": eo.masked_marginals_table(ocfg, W, seq, positions=list(positions), batch=batch)}
        # (tests/frozen.py: the oracle's rows are a function of fixed seeds; the stored copy is used while everything still hashes the same)
        _ORACLE[key] = frozen.cached(f"baseline_{name}_std{embed_std}_seed{seed}", [name, embed_std, seed, seq, list(positions), batch, blob], compute)["table"]
    return _ORACLE[key]


def _score_oracle(muts, seq, table):
    from oracle import esm_oracle as eo
    return np.array([eo.label_row(m, seq, table, 1) for m in muts])


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_esm1v_650m_blat_full_assay_vs_oracle(lib, precision):
    """BASELINE config 2 at its own shape: all rows of the assay's table and every score, flat 1e-4."""
    name, std, seed = "ESM1V_650M", 0.15, 1
    cfg, blob = _weights(name, std, seed)
    seq, muts, _ = synthetic.random_assay(seed=23, L=L_BLAT, n_single=4996, n_multi=400)
    model = pesm.EsmModel(cfg, blob, device=0, precision=precision)
    assay = pesm.Assay(model, seq, muts)
    scores, table = assay.run(want_table=True)
    positions = [int(p) for p in assay.positions]
    assert len(positions) >= 280                                     # 4 996 singles touch (nearly) every residue
    ref = _oracle_table(name, std, seed, seq, positions, batch=16)
    err_t = float(np.abs(table[positions] - ref[positions]).max())
    ref_s = _score_oracle(muts, seq, ref)
    err_s = np.abs(scores - ref_s)
    depth = np.array([m.count(":") + 1 for m in muts])
    llr = ref[positions][:, 4:24]
    print(f"[{precision}] ESM-1v 650M T=288: {len(positions)} table rows max|err| {err_t:.2e}; scores: singles "
          f"{err_s[depth == 1].max():.2e}, multi (depth 2-5, n={int((depth > 1).sum())}) {err_s[depth > 1].max():.2e}, "
          f"depth-5 {err_s[depth == 5].max():.2e}; log-prob range {float(llr.max() - llr.min()):.1f}, "
          f"|score| max {np.abs(ref_s).max():.1f}")
    assert err_t < TOL
    assert err_s.max() < TOL
    # Spearman vs a synthetic DMS_score identical to 4 dp with the oracle's (the north-star's second clause)
    from scipy.stats import spearmanr
    dms = np.random.default_rng(0).standard_normal(len(muts)) + 0.3 * ref_s
    assert round(spearmanr(scores, dms)[0], 4) == round(spearmanr(ref_s, dms)[0], 4)
    assay.close()
    model.close()


def _esm2_3b_case(precision, embed_std):
    name, seed = "ESM2_3B", 9
    cfg, blob = _weights(name, embed_std, seed)
    rng = np.random.default_rng(4)
    seq = synthetic.random_sequence(rng, L_BLAT)
    aa = list(synthetic.AA)

    def sub(p):
        return f"{seq[p]}{p + 1}{rng.choice([a for a in aa if a != seq[p]])}"
    d5a = sorted(rng.choice(L_BLAT, 5, replace=False).tolist())
    d5b = sorted(rng.choice(L_BLAT, 5, replace=False).tolist())
    extra = sorted(rng.choice(L_BLAT, 6, replace=False).tolist())
    muts = [":".join(sub(p) for p in d5a), ":".join(sub(p) for p in d5b),
            ":".join(sub(p) for p in d5a[:3]), ":".join(sub(p) for p in d5b[1:])]
    for p in d5a + d5b + extra + [0, L_BLAT - 1]:
        muts += [sub(p), sub(p)]
    model = pesm.EsmModel(cfg, blob, device=0, precision=precision)
    assay = pesm.Assay(model, seq, muts)
    scores, table = assay.run(want_table=True)
    positions = [int(p) for p in assay.positions]
    ref = _oracle_table(name, embed_std, seed, seq, positions, batch=6)
    err_t = float(np.abs(table[positions] - ref[positions]).max())
    ref_s = _score_oracle(muts, seq, ref)
    err_s = np.abs(scores - ref_s)
    llr = ref[positions][:, 4:24]
    rng_lp = float(llr.max() - llr.min())
    print(f"[{precision}] ESM2-3B FULL DEPTH (36x2560x40) T=288 embed_std={embed_std}: {len(positions)} rows max|err| "
          f"{err_t:.2e}; score max|err| {err_s.max():.2e} (depth-5: {err_s[:2].max():.2e}); log-prob range {rng_lp:.1f}")
    assay.close()
    model.close()
    return err_t, float(err_s.max()), rng_lp


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_esm2_3b_full_depth_vs_oracle(lib, precision):
    """BASELINE config 3's model at full depth, realistic-range weights: flat 1e-4 on rows and scores."""
    err_t, err_s, _ = _esm2_3b_case(precision, 0.15)
    assert err_t < TOL
    assert err_s < TOL


def test_esm2_3b_full_depth_stress_reported(lib):
    """Default-init-like embeddings (N(0,1): log-prob range several times a real checkpoint's): reported, and held to
    a RELATIVE bar of 1e-5 x range (at such ranges the reference's own fp32 rounding is already
    ~1e-4; measured here: 5.7e-4 at a range of 190 = 3e-6 x range, profiles/r2/parity_baseline_shapes.log)."""
    err_t, err_s, rng_lp = _esm2_3b_case("f16x3", 1.0)
    assert err_t < max(TOL, 1e-5 * rng_lp)
    assert err_s < max(TOL, 5e-5 * rng_lp)
