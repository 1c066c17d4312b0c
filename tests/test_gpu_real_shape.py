"""Parity at the real ESM-1v 650M shape (33 x 1280 x 20 x 5120, synthetic weights): HIP path
through the C ABI vs the oracle (CPU fp32) on seeded inputs, plus size-independent properties at
the full BLAT-shaped workload."""
import numpy as np
import pytest

from proteingym_amd import esm as pesm, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fp32", "f16x3"])
def big(lib, request):
    cfg = dict(synthetic.ESM1V_650M)
    blob = _blob()
    model = pesm.EsmModel(cfg, blob, device=0, precision=request.param)
    yield cfg, blob, model
    model.close()


_BLOB = {}


def _blob():
    if "b" not in _BLOB:
        _BLOB["b"] = synthetic.random_weights(dict(synthetic.ESM1V_650M), seed=1)
    return _BLOB["b"]


_REF = {}


def _oracle_rows(cfg, blob, seq, positions):
    """CPU oracle rows at the real shape, computed once and shared by the precision modes."""
    from oracle import esm_oracle as eo
    key = (seq, tuple(positions))
    if key not in _REF:
        ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), **cfg)
        _REF[key] = eo.masked_marginals_table(ocfg, W, seq, positions=positions, batch=len(positions))
    return _REF[key]


def test_650m_masked_rows_vs_oracle(big):
    from oracle import esm_oracle as eo
    cfg, blob, model = big
    seq, muts, _ = synthetic.random_assay(seed=5, L=120, n_single=200, n_multi=50)
    positions = [1, 17, 60, 119, 120]
    ref = _oracle_rows(cfg, blob, seq, positions)
    toks = eo.tokenize(seq)
    got = model.masked_logprobs(np.repeat(toks[None], len(positions), 0), positions)
    err = np.abs(got - ref[positions]).max()
    llr = ref[positions][:, 4:24]
    print(model.precision, "650M max|err| =", err, " LLR range", float(llr.max() - llr.min()))
    assert err < 1e-4


def test_650m_full_assay_properties(big):
    """BLAT-shaped workload (L=286, 4996 mutants): properties that need no oracle."""
    cfg, blob, model = big
    seq, muts, _ = synthetic.random_assay(seed=23, L=286, n_single=4996, n_multi=0)
    a = pesm.Assay(model, seq, muts)
    s1, table = a.run(want_table=True)
    s2 = a.run()
    assert np.array_equal(s1, s2)                                     # deterministic
    done = ~np.isnan(table[:, 0])
    assert np.allclose(np.exp(table[done].astype(np.float64)).sum(-1), 1.0, atol=1e-5)   # rows are log-probs
    # score == table lookup (label_row) recomputed on the host from the returned table
    sub_pos, sub_wt, sub_mt, off = pesm.parse_mutants(muts, seq, 1)
    host = np.array([np.sum((table[sub_pos[off[i]:off[i + 1]], sub_mt[off[i]:off[i + 1]]]
                             - table[sub_pos[off[i]:off[i + 1]], sub_wt[off[i]:off[i + 1]]]).astype(np.float64))
                     for i in range(len(muts))])
    assert np.array_equal(host, s1)
    # a multi-mutant is the sum of its single-site terms (linearity of masked-marginals)
    m = [muts[0], muts[1], muts[0] + ":" + muts[1]] if muts[0][1:-1] != muts[1][1:-1] else None
    if m:
        b = pesm.Assay(model, seq, m).run()
        assert abs(b[2] - (b[0] + b[1])) < 1e-12


def test_tranception_large_shape_vs_oracle(lib):
    """Tranception-L shape (36 x 1280, 20 heads -> grouped ALiBi over 5 slopes, FFN 5120), synthetic
    weights: per-sequence log-likelihoods and token log-probs of the HIP path vs the CPU oracle."""
    from oracle import tranception_oracle as to
    from proteingym_amd import tranception as ptr
    cfg = dict(synthetic.TRANCEPTION_L)
    blob = synthetic.random_tranception_weights(cfg, seed=3)
    model = ptr.TranceptionModel(cfg, blob, device=0)
    ocfg, W = to.from_arrays(arrays=synthetic.tranception_blob_to_arrays(cfg, blob), **cfg)
    rng = np.random.default_rng(0)
    seqs = ["".join(rng.choice(list(synthetic.AA), size=n)) for n in (90, 90, 61, 33)]
    ids, mask = to.encode_batch(seqs[:2])
    import torch
    with torch.no_grad():
        lg32 = to.forward_logits(ocfg, W, ids, mask)
        ref = torch.log_softmax(lg32, -1).numpy()
        W64 = {k: v.double() for k, v in W.items()}
        ref64 = torch.log_softmax(to.forward_logits(ocfg, W64, ids, mask), -1).numpy()
    got = model.token_logprobs(ids)
    noise = np.abs(ref - ref64).max()                  # the fp32 CPU path's own distance to fp64
    err64 = np.abs(got - ref64).max()
    print("logit range", float(lg32.max() - lg32.min()), " fp32-oracle vs fp64:", noise, " HIP vs fp64:", err64)
    assert err64 < max(1e-4, 3.0 * noise)             # same class as the reference's own fp32 arithmetic (token level)
    err = np.abs(got - ref).max()
    ref_ll = to.sequence_scores(ocfg, W, seqs, [0] * 4, [len(s) for s in seqs])
    got_ll = model.sequence_loglik(seqs)
    print("Tranception-L token log-prob max|err| =", err, " per-residue LL err =",
          np.abs((got_ll - ref_ll) / np.array([len(s) for s in seqs])).max())
    assert err < 1e-4 + 3.0 * noise
    assert np.abs((got_ll - ref_ll) / np.array([len(s) for s in seqs])).max() < 1e-4     # the scored quantity
    model.close()


@pytest.mark.parametrize("name,layers", [("ESM2_3B", 3), ("ESM2_650M", 6)])
def test_esm2_widths_vs_oracle(lib, name, layers):
    """ESM2 (rotary, no learned positions) at the 3B width (2560, 40 heads, FFN 10240: BASELINE config 3) and the
    650M width, truncated in depth so the CPU oracle stays in seconds; default precision (f16x3).  The synthetic
    stress weights give a log-prob range of ~55 at the 3B width (real checkpoints: < 20), where the reference's own
    fp32 arithmetic is 5e-5 away from fp64; the bar is 1e-4 abs against the fp64 oracle or 3x that fp32 noise,
    whichever is larger (same rule as the Tranception-L test)."""
    import torch
    from oracle import esm_oracle as eo
    cfg = dict(getattr(synthetic, name), layers=layers)
    blob = synthetic.random_weights(cfg, seed=9)
    seq, muts, _ = synthetic.random_assay(seed=2, L=75, n_single=60, n_multi=20)
    model = pesm.EsmModel(cfg, blob, device=0)
    a = pesm.Assay(model, seq, muts)
    scores, table = a.run(want_table=True)
    arrays = synthetic.blob_to_arrays(cfg, blob)
    positions = sorted(int(p) for p in a.positions)
    tabs = {}
    for dt in (torch.float32, torch.float64):
        ocfg, W = eo.from_arrays(arrays=arrays, dtype=dt, **cfg)
        tabs[dt] = eo.masked_marginals_table(ocfg, W, seq, positions=positions, batch=16)
    t32, t64 = tabs[torch.float32], tabs[torch.float64]
    noise = float(np.abs(t32[positions] - t64[positions]).max())
    err = float(np.abs(table[positions] - t64[positions]).max())
    s32 = np.array([eo.label_row(m, seq, t32, 1) for m in muts])
    s64 = np.array([eo.label_row(m, seq, t64, 1) for m in muts])
    s_noise, s_err = float(np.abs(s32 - s64).max()), float(np.abs(scores - s64).max())
    print(f"{name}: table err {err:.2e} (fp32 reference noise {noise:.2e}), score err {s_err:.2e} (noise {s_noise:.2e})")
    assert err < max(1e-4, 3 * noise)
    assert s_err < max(1e-4, 3 * s_noise)
    a.close()
    model.close()
