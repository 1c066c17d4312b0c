"""Parity at real model WIDTHS on the configurations the shorter tests never reach:

  * ESM-1v 650M through the 1 024-token window (16 of the 217 substitution assays are longer than 1 022 residues:
    every one of their forwards runs T = 1024 -- 32 key tiles per online softmax, learned positions up to 1 025);
  * ESM2-650M pseudo-perplexity at BASELINE config 5's own shape (two ~735-residue members, 736 / 735-term sums) against
    terms the UNMODIFIED reference model produced (tests/golden/make_golden_real_width.py), with the reference arithmetic's own
    distance to fp64 frozen next to them;
  * Tranception-L shape at its full context (n_ctx 1024: 32 causal key tiles, ALiBi bias up to key 1 023), both reading
    directions, and a 1 100-residue protein through the optimal-window scorer;
  * ESM2-15B layer shape (5120 wide, 40 heads of 128, FFN 20480) at 8 layers.

Bars: flat 1e-4 abs against the CPU fp32 oracle / the reference goldens (the north-star's bar) unless a comment says why
not; where the fp32 reference's own distance to exact arithmetic is the same size as the bar, the fp64 oracle is the
yardstick and the fp32 noise is printed next to the result.
References: /root/reference/proteingym/baselines/esm/compute_fitness.py:258-279,486-504; proteingym/utils/scoring_utils.py:43-52;
proteingym/baselines/tranception/tranception/model_pytorch.py:155-183,878-928, utils/scoring_utils.py:152-203.
"""
import os

import numpy as np
import pytest

from proteingym_amd import esm as pesm, synthetic

import frozen  # noqa: E402  (tests/frozen.py)

pytestmark = pytest.mark.gpu
TOL = 1e-4

_BLOBS = {}


def _blob(name, seed, embed_std, layers=None):
    key = (name, seed, embed_std, layers)
    if key not in _BLOBS:
        _BLOBS.clear()                                                   # one multi-GB blob at a time
        cfg = dict(getattr(synthetic, name))
        if layers:
            cfg["layers"] = layers
        _BLOBS[key] = (cfg, synthetic.random_weights(cfg, seed=seed, embed_std=embed_std))
    return _BLOBS[key]


def _threads():
    import torch
    torch.set_num_threads(max(1, __import__("bench").usable_cores()))


# ---------------------------------------------------------------------------------------------------------------------
# (i) ESM-1v 650M, a 1 100-residue protein: every forward is a 1 024-token window
# ---------------------------------------------------------------------------------------------------------------------
_LONG = {}


def _long_case():
    """Sequence, mutants and the oracle rows (computed once, shared by the precision modes)."""
    if "ref" in _LONG:
        return _LONG["seq"], _LONG["muts"], _LONG["positions"], _LONG["ref"]
    from oracle import esm_oracle as eo
    rng = np.random.default_rng(1100)
    L = 1100
    seq = synthetic.random_sequence(rng, L)
    aa = list(synthetic.AA)

    def sub(p):                                                          # residue index p (0-based) -> "A17G"
        return f"{seq[p]}{p + 1}{rng.choice([a for a in aa if a != seq[p]])}"
    # token i = p + 1; n = 1102 tokens, W = 1024: windows [0,1024) for i < 512, [i-512, i+512) for 512 <= i < 590,
    # [78, 1102) for i >= 590 (scoring_utils.py:43-52).  Both window edges, both ends of the protein, the middle.
    residues = [0, 510, 511, 550, 588, 589, 1023, 1099]
    muts = [sub(p) for p in residues]
    muts.append(":".join(sub(p) for p in (0, 511, 550, 589, 1099)))      # depth 5 across all three window kinds
    muts.append(":".join(sub(p) for p in (510, 588)))
    positions = sorted(p + 1 for p in residues)
    cfg, blob = _blob("ESM1V_650M", 1, 0.15)

    def compute():
        _threads()
        ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), **cfg)
        return {"table": eo.masked_marginals_table(ocfg, W, seq, positions=positions, batch=4)}
    ref = frozen.cached("real_width_esm1v_650m_L1100_windows", ["ESM1V_650M", 1, 0.15, seq, positions, blob], compute)["table"]
    _LONG.update(seq=seq, muts=muts, positions=positions, ref=ref)
    return seq, muts, positions, ref


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_esm1v_650m_t1024_windows_vs_oracle(lib, precision):
    from oracle import esm_oracle as eo
    seq, muts, positions, ref = _long_case()
    cfg, blob = _blob("ESM1V_650M", 1, 0.15)
    model = pesm.EsmModel(cfg, blob, device=0, precision=precision)
    assay = pesm.Assay(model, seq, muts)
    assert assay.T == 1024 and sorted(int(p) for p in assay.positions) == positions
    scores, table = assay.run(want_table=True)
    err_t = float(np.abs(table[positions] - ref[positions]).max())
    ref_s = np.array([eo.label_row(m, seq, ref, 1) for m in muts])
    err_s = np.abs(scores - ref_s)
    llr = ref[positions][:, 4:24]
    print(f"[{precision}] ESM-1v 650M, L=1100 (T=1024 windows): {len(positions)} rows max|err| {err_t:.2e}; scores max|err| "
          f"{err_s.max():.2e} (depth-5 {err_s[-2]:.2e}); log-prob range {float(llr.max() - llr.min()):.1f}")
    assert err_t < TOL
    assert err_s.max() < TOL
    assay.close()
    model.close()


# ---------------------------------------------------------------------------------------------------------------------
# (ii) ESM2-650M pseudo-ppl at config 5's shape vs reference-generated goldens
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_esm2_650m_pppl_735_residues_vs_reference(lib, golden_dir, precision):
    path = os.path.join(golden_dir, "golden_pppl_650m.npz")
    if not os.path.exists(path):
        pytest.skip("golden_pppl_650m.npz not generated (tests/golden/make_golden_real_width.py)")
    g = np.load(path)
    seqs = [str(g[f"seq/{r}"]) for r in range(2) if f"terms/{r}" in g]
    assert seqs == synthetic.random_indel_library(7, 735, 2)[1][: len(seqs)]        # the fixture's inputs are reproducible
    cfg, blob = _blob("ESM2_650M", 5, 0.15)
    model = pesm.EsmModel(cfg, blob, device=0, precision=precision)
    lib_ = pesm.SequenceLibrary(model, seqs)
    scores, terms = lib_.score(want_terms=True)
    # The sum bar.  Config 5's per-mutant quantity is a SUM of ~736 terms, and the north star's flat 1e-4 cannot be stated against the
    # reference for it: the UNMODIFIED reference's own fp32 sum is 2.6e-3 away from the same sum in exact (fp64) arithmetic (terms64/0
    # of the fixture: per-term 2.4e-5; the per-term differences do not average out over the sum).  What is held, in BOTH modes:
    #   * every TERM within the flat 1e-4 of the reference's;
    #   * the sum CLOSER to exact (fp64) arithmetic than the reference's own sum is (e64 < the reference's distance), hence
    #   * |sum - reference's sum| < 2 x that distance (triangle inequality) -- i.e. "within the reference's own fp32 noise", not 1e-4.
    # Measured (profiles/r4/README.md): f16x3 1.4e-3 from fp64 (an fp16 MFMA adds 16 exact products before its one fp32 rounding), the
    # fp32 mode 3.2e-4 since its GEMM sums per K tile first (round 3, one chain over K: 5.7e-3 = 2.2x the reference's distance).
    ref_noise = abs(float(g["sum/0"]) - g["terms64/0"].sum()) if "terms64/0" in g else 2.6e-3
    for r, s in enumerate(seqs):
        ref = g[f"terms/{r}"]
        assert len(terms[r]) == len(ref) == len(s) - 2
        e_term = float(np.abs(terms[r].astype(np.float64) - ref).max())
        e_sum = abs(scores[r] - float(g[f"sum/{r}"]))
        msg = (f"[{precision}] ESM2-650M pseudo-ppl, member {r} ({len(s)} residues, {len(ref)} terms): per-term max|err| {e_term:.2e}; "
               f"sum |err| {e_sum:.2e} on {float(g[f'sum/{r}']):.2f} (relative {e_sum / abs(float(g[f'sum/{r}'])):.1e})")
        if f"terms64/{r}" in g:
            t64 = g[f"terms64/{r}"]
            e64 = abs(scores[r] - t64.sum())
            msg += (f"; the reference's own fp32 arithmetic vs fp64: per-term {np.abs(ref - t64).max():.2e}, sum {abs(float(g[f'sum/{r}']) - t64.sum()):.2e}"
                    f"; HIP vs fp64: per-term {np.abs(terms[r] - t64).max():.2e}, sum {e64:.2e}")
            assert e64 < ref_noise
        print(msg)
        assert scores[r] == sum(float(v) for v in terms[r])                          # python's left-to-right double sum of the f32 terms
        assert e_term < TOL                                                          # flat 1e-4 on every term
        assert e_sum < max(TOL, 2.0 * ref_noise)
    # a member scored alone (another batch composition: what a rank of run_indels sees) has the SAME BITS
    for r in range(len(seqs)):
        alone, t_alone = lib_.score(first=r, count=1, want_terms=True)
        assert np.array_equal(t_alone[0], terms[r])
        assert alone[0] == scores[r]
    lib_.close()
    model.close()


# ---------------------------------------------------------------------------------------------------------------------
# (iii) Tranception-L shape at the full 1 024-token context
# ---------------------------------------------------------------------------------------------------------------------
def test_tranception_l_full_context_vs_oracle(lib):
    import torch
    import pandas as pd
    from oracle import tranception_oracle as to
    from proteingym_amd import tranception as ptr
    _threads()
    cfg = dict(synthetic.TRANCEPTION_L)
    blob = synthetic.random_tranception_weights(cfg, seed=3)
    model = ptr.TranceptionModel(cfg, blob, device=0)
    ocfg, W = to.from_arrays(arrays=synthetic.tranception_blob_to_arrays(cfg, blob), **cfg)
    rng = np.random.default_rng(1024)
    full = "".join(rng.choice(list(synthetic.AA), size=1022))              # [CLS] + 1022 + [SEP] = n_ctx tokens
    seqs = [full, full[::-1]]                                              # the scorer's two reading directions
    ids, mask = to.encode_batch(seqs)
    assert ids.shape == (2, 1024)

    def compute_ctx():
        with torch.no_grad():
            r32 = torch.log_softmax(to.forward_logits(ocfg, W, ids, mask), -1).numpy()
            W64 = {k: v.double() for k, v in W.items()}
            r64 = torch.log_softmax(to.forward_logits(ocfg, W64, ids, mask), -1).numpy()
        return {"ref": r32, "ref64": r64}
    fz = frozen.cached("real_width_tranception_l_n_ctx_1024", ["TRANCEPTION_L", 3, seqs, np.asarray(ids), blob], compute_ctx)
    ref, ref64 = fz["ref"], fz["ref64"]
    got = model.token_logprobs(ids)
    noise = float(np.abs(ref - ref64).max())
    err64 = float(np.abs(got - ref64).max())
    err32 = float(np.abs(got - ref).max())
    # the scored quantity: log p(token t+1 | <= t) summed over the sequence, per residue (scoring_utils.py:118-141)
    tgt = ids[:, 1:]
    pick = lambda lp: np.take_along_axis(lp[:, :-1], tgt[..., None], -1)[..., 0]
    tok_err = float(np.abs(pick(got) - pick(ref64)).max())
    ll = model.sequence_loglik(seqs)
    ll_ref = pick(ref64).sum(1)
    ll_err = float(np.abs(ll - ll_ref).max() / 1022)
    print(f"Tranception-L at n_ctx 1024: token log-probs HIP vs fp64 {err64:.2e} (all 25 symbols) / {tok_err:.2e} (the scored symbol), "
          f"HIP vs CPU fp32 {err32:.2e}, CPU fp32 vs fp64 {noise:.2e}; per-residue log-likelihood |err| {ll_err:.2e} "
          f"(sum {float(ll_ref[0]):.1f})")
    assert err64 < max(TOL, 3.0 * noise)
    assert tok_err < max(TOL, 3.0 * noise)
    assert ll_err < TOL
    # a 1 100-residue protein through score_mutants (optimal windows of 1 022 residues, both directions, delta to the wild
    # type of the same window) against the oracle's scorer
    L = 1100
    wt = "".join(rng.choice(list(synthetic.AA), size=L))
    muts = []
    for p in (3, 511, 700, 1096):
        muts.append(f"{wt[p]}{p + 1}{'A' if wt[p] != 'A' else 'C'}")
    df = pd.DataFrame({"mutant": muts, "mutated_sequence": [ptr.get_mutated_sequence(wt, m) for m in muts]})   # as the real DMS files
    cols = ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score")

    def compute_scores():
        with torch.no_grad():
            w = to.score_mutants(ocfg, W, df, wt)
        return {"mutated_sequence": np.array(list(w["mutated_sequence"])), **{c: w[c].to_numpy() for c in cols}}
    want = frozen.cached("real_width_tranception_l_L1100_scores", ["TRANCEPTION_L", 3, wt, muts, blob], compute_scores)
    have = model.score_mutants(DMS_data=df, target_seq=wt)
    assert list(have["mutated_sequence"]) == [str(x) for x in want["mutated_sequence"]]
    for col in cols:
        e = float(np.abs(have[col].to_numpy() - want[col]).max())
        print(f"Tranception-L, L=1100 optimal windows: {col} max|err| {e:.2e}")
        assert e < TOL
    model.close()


# ---------------------------------------------------------------------------------------------------------------------
# (iv) ESM2-15B layer shape at 8 layers
# ---------------------------------------------------------------------------------------------------------------------
def _avail_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except Exception:
        pass
    return 0.0


def test_esm2_15b_width_8_layers_vs_oracle(lib):
    """Operand rounding at K = 5120 / 20480 accumulates over depth: 8 of the 48 layers (2.5 G parameters, 10 GB of fp32
    weights), T = 152, realistic-range weights (embed_std 0.075, see test_esm2_15b_width_vs_oracle)."""
    import torch
    from oracle import esm_oracle as eo
    layers = int(os.environ.get("PGMI_TEST_15B_LAYERS", "8"))
    need = 1.3 * layers + 4 if os.path.exists(os.path.join(frozen.FROZEN_DIR, f"real_width_esm2_15b_{layers}_layers.npz")) \
        else 3.3 * layers + 8                                            # blob (+ fp32 oracle, which shares its memory, + fp64 copy when the oracle runs live)
    if _avail_gb() < need:
        pytest.skip(f"needs ~{need:.0f} GB of host memory for the fp64 oracle")
    _threads()
    cfg, blob = _blob("ESM2_15B", 15, 0.075, layers=layers)
    seq, muts, _ = synthetic.random_assay(seed=8, L=150, n_single=40, n_multi=10)
    m = pesm.EsmModel(cfg, blob, device=0)
    a = pesm.Assay(m, seq, muts)
    scores, table = a.run(want_table=True)
    a.close()
    m.close()
    positions = sorted(int(p) for p in a.positions)[:12]

    def compute():
        tabs = {}
        for tag, dt in (("t64", torch.float64), ("t32", torch.float32)):
            ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), dtype=dt, **cfg)
            tabs[tag] = eo.masked_marginals_table(ocfg, W, seq, positions=positions, batch=6)
            del W
        return tabs
    fz = frozen.cached(f"real_width_esm2_15b_{layers}_layers", ["ESM2_15B", 15, 0.075, layers, seq, positions, blob], compute)
    t32, t64 = fz["t32"], fz["t64"]
    noise = float(np.abs(t32[positions] - t64[positions]).max())
    err64 = float(np.abs(table[positions] - t64[positions]).max())
    err32 = float(np.abs(table[positions] - t32[positions]).max())
    rng_lp = float(t64[positions][:, 4:24].max() - t64[positions][:, 4:24].min())
    print(f"ESM2-15B width, {layers} layers: HIP vs fp64 {err64:.2e}, HIP vs CPU fp32 {err32:.2e}, CPU fp32 vs fp64 {noise:.2e}, "
          f"log-prob range {rng_lp:.1f}")
    assert err64 < TOL
    assert err32 < TOL + noise


@pytest.mark.skipif(os.environ.get("PGMI_TEST_15B_FULL") == "0", reason="ESM2-15B at full depth (60 GB of weights, ~30 s, needs 200 GB of free host memory) switched off by PGMI_TEST_15B_FULL=0")
def test_esm2_15b_full_depth_equals_its_8_layer_prefix(lib):
    """ESM2-15B as released (48 x 5120, 40 heads of 128, FFN 20480: 15.1 G parameters, 60 GB of split weight planes resident; launcher:
    scripts/scoring_DMS_zero_shot/scoring_ESM2_substitutions.sh) instantiated ONCE at full depth.  Layers 0 - 7 are the 8-layer model of
    test_esm2_15b_width_8_layers_vs_oracle (4.2e-5 against the fp64 oracle); layers 8 - 47 carry layer 7's LayerNorms, q / k / v and
    FC1 (the same compute as any layer) and ZERO out-projection and FC2, so each of them adds exactly 0 to the residual stream: the
    48-layer table must equal the 8-layer table bit for bit -- through 40 more layers of kernels at the 15B shape, the last layer's
    kept-rows path included -- while the memory plan (weights + workspace) and the launch sequence are those of the real model."""
    import time
    if _avail_gb() < 200:
        pytest.skip("needs ~130 GB of host memory for the 60 GB blob and its source")
    cfg8, blob8 = _blob("ESM2_15B", 15, 0.075, layers=8)
    seq, muts, _ = synthetic.random_assay(seed=8, L=150, n_single=40, n_multi=10)
    m = pesm.EsmModel(cfg8, blob8, device=0, max_rows=16384)
    a = pesm.Assay(m, seq, muts)
    s8, t8 = a.run(want_table=True)
    a.close()
    m.close()
    cfg = dict(synthetic.ESM2_15B)
    shapes = synthetic.key_shapes(cfg)
    blob = np.zeros(sum(int(np.prod(sh)) for _, sh in shapes), dtype=np.float32)
    src, dst = synthetic.blob_to_arrays(cfg8, blob8), synthetic.blob_to_arrays(cfg, blob)
    for k, v in dst.items():
        if k == "lm_head.weight":
            continue
        if not k.startswith("layers."):
            v[...] = src[k]
            continue
        layer, leaf = int(k.split(".")[1]), k.split(".", 2)[2]
        if layer < 8:
            v[...] = src[k]
        elif not (leaf.startswith("self_attn.out_proj") or leaf.startswith("fc2")):
            v[...] = src[f"layers.7.{leaf}"]
    t0 = time.time()
    m = pesm.EsmModel(cfg, blob, device=0, max_rows=16384)
    t_create = time.time() - t0
    a = pesm.Assay(m, seq, muts)
    t0 = time.time()
    s48, t48 = a.run(want_table=True)
    t_run = time.time() - t0
    a.close()
    m.close()
    print(f"ESM2-15B, 48 layers ({blob.size / 1e9:.1f} G parameters): model created in {t_create:.0f} s, {len(a.positions)} masked forwards of {a.T} tokens in "
          f"{t_run:.2f} s; table and scores equal the 8-layer prefix bit for bit")
    assert np.isfinite(s48).all()
    assert np.array_equal(t48, t8, equal_nan=True) and np.array_equal(s48, s8)


def test_esm1v_650m_short_assays_grouped_equal_one_at_a_time(lib):
    """run_benchmark's short-assay groups at the REAL width (33 x 1280 x 20): three proteins of 40 / 57 / 95 residues, their masked
    copies in one padded launch sequence, against one Assay.run() per protein -- every score bit for bit."""
    from proteingym_amd import run_benchmark as rb
    cfg = dict(synthetic.ESM1V_650M)
    model = pesm.EsmModel(cfg, synthetic.random_weights(cfg, seed=1, embed_std=0.15), device=0, precision="f16x3")
    sc = rb._DeviceScorer.__new__(rb._DeviceScorer)
    sc.model, sc.alphabet, sc.all_positions, sc.log, sc.create_s, sc.run_s = model, pesm.Alphabet(), False, [], 0.0, 0.0
    assays = []
    for k, L in enumerate((40, 57, 95)):
        seq, muts, _ = synthetic.random_assay(seed=70 + k, L=L, n_single=2 * L, n_multi=20)
        assays.append((seq, muts, 1))
    grouped = sc.score_group(assays)
    for (seq, muts, off), g in zip(assays, grouped):
        one = sc.score(seq, muts, off)
        assert np.array_equal(np.asarray(g), np.asarray(one)), f"L={len(seq)}: max |diff| {np.abs(np.asarray(g) - np.asarray(one)).max():.3e}"
        assert np.isfinite(one).all()
    print(f"ESM-1v 650M short-assay group (40/57/95 residues, padded to 97 tokens): scores bit-identical to one assay at a time; "
          f"group {sc.log[0]['run_s'] + sc.log[1]['run_s'] + sc.log[2]['run_s']:.3f} s vs one at a time {sum(e['run_s'] for e in sc.log[3:]):.3f} s")
    model.close()
