"""GPU parity of the HIP path (through the C ABI) against the reference's own outputs frozen in
tests/golden/golden_esm.npz, and against the oracle on seeded inputs."""
import os

import numpy as np
import pytest

from proteingym_amd import esm as pesm, synthetic

pytestmark = pytest.mark.gpu

TOL = 1e-4   # BASELINE.json north_star: per-mutant scores within 1e-4 abs of the reference CPU path


@pytest.fixture(scope="module", params=["f16x3", "fp32"])
def models(lib, golden_dir, request):
    """Both parity-gated precision modes go through every test below."""
    out = {}
    for n in ("esm1v_toy_1", "esm1v_toy_2", "esm1b_toy_lnb", "esm2_toy"):
        out[n] = pesm.load_model_and_alphabet(os.path.join(golden_dir, n + ".pt"), precision=request.param)[0]
    yield out
    for m in out.values():
        m.close()


@pytest.mark.parametrize("name", ["esm1v_toy_1", "esm1b_toy_lnb", "esm2_toy"])
def test_wt_logprobs_vs_reference(models, golden, name):
    seq = str(golden["seq"])
    _, _, toks = pesm.Alphabet().get_batch_converter()([("p", seq)])
    lp = models[name](toks)["logits"][0]
    ref = golden[f"{name}/wt_logprobs"]
    assert np.abs(lp - ref).max() < TOL


@pytest.mark.parametrize("name", ["esm1v_toy_1", "esm1b_toy_lnb", "esm2_toy"])
def test_padded_batch_vs_reference(models, golden, name):
    toks = golden[f"{name}/pad_tokens"]
    ref = golden[f"{name}/pad_logprobs"]
    lp = models[name].token_logprobs(toks)
    valid = toks != 1
    assert np.abs(lp[valid] - ref[valid]).max() < TOL


@pytest.mark.parametrize("name", ["esm1v_toy_1", "esm1v_toy_2", "esm1b_toy_lnb", "esm2_toy"])
def test_masked_marginals_table_vs_reference(models, golden, name):
    seq = str(golden["seq"])
    _, _, toks = pesm.Alphabet().get_batch_converter()([("p", seq)])
    n = toks.shape[1]
    lp = models[name].masked_logprobs(np.repeat(toks, n, axis=0), np.arange(n))
    assert np.abs(lp - golden[f"{name}/mm_table"]).max() < TOL


def test_assay_scores_vs_reference_cli(models, golden, golden_dir):
    import pandas as pd
    seq = str(golden["seq"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    cols = {}
    for name in ("esm1v_toy_1", "esm1v_toy_2", "esm2_toy", "esm1b_toy_lnb"):
        a = pesm.Assay(models[name], seq, list(df["mutant"]), offset_idx=1)
        scores, table = a.run(want_table=True)
        cols[name] = scores
        assert np.abs(scores - golden[f"cli/{name}"]).max() < TOL
        # positions no mutant touches are skipped (NaN), the others match the reference table
        ref_t = golden[f"{name}/mm_table"]
        done = ~np.isnan(table[:, 0])
        assert done.sum() == len(a.positions)
        assert np.abs(table[done] - ref_t[done]).max() < TOL
        a.close()
    ens = (cols["esm1v_toy_1"] + cols["esm1v_toy_2"]) / 2
    assert np.abs(ens - golden["cli/Ensemble_ESM1v"]).max() < TOL


def test_all_positions_equals_subset(models, golden, golden_dir):
    import pandas as pd
    seq = str(golden["seq"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    m = models["esm1v_toy_1"]
    a = pesm.Assay(m, seq, list(df["mutant"]), all_positions=True)
    b = pesm.Assay(m, seq, list(df["mutant"]), all_positions=False)
    sa, ta = a.run(want_table=True)
    sb = b.run()
    assert np.array_equal(sa, sb)            # bit-identical: per-position results do not depend on batch
    # pgmi_score_mutants (host C, the table -> scores export) gives the device's score_mutants_kernel bits
    assert np.array_equal(pesm.score_from_table(ta, list(df["mutant"]), seq, 1), sa)
    assert not np.isnan(ta).any()
    assert np.abs(ta - golden["esm1v_toy_1/mm_table"]).max() < TOL


@pytest.mark.parametrize("name", ["esm1v_toy_1", "esm2_toy"])
def test_long_protein_optimal_window(models, golden, golden_dir, name):
    import pandas as pd
    seq = str(golden["seq_long"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_LONG_DMS.csv"))
    a = pesm.Assay(models[name], seq, list(df["mutant"]))
    scores = a.run()
    assert np.abs(scores - golden[f"cli_long/{name}"]).max() < TOL


def test_determinism_and_chunking(lib, golden, golden_dir):
    """Same input -> bit-identical output across runs and across internal chunk sizes."""
    import pandas as pd
    seq = str(golden["seq"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    path = os.path.join(golden_dir, "esm1v_toy_1.pt")
    m1 = pesm.load_model_and_alphabet(path, max_rows=2048)[0]     # forces several chunks
    m2 = pesm.load_model_and_alphabet(path)[0]
    s1 = pesm.Assay(m1, seq, list(df["mutant"])).run()
    s1b = pesm.Assay(m1, seq, list(df["mutant"])).run()
    s2 = pesm.Assay(m2, seq, list(df["mutant"])).run()
    assert np.array_equal(s1, s1b)
    assert np.array_equal(s1, s2)


@pytest.mark.parametrize("name", ["esm1v_toy_1", "esm2_toy", "esm2_toy_h128"])
def test_gemm_row_chunks_inside_a_model_bit_identical(lib, golden, golden_dir, name, gemm_option):
    """Every GEMM of the forward cut into row chunks (the path activations beyond 4 GiB take: gemm_f16.hip launch_gemm16x;
    the fused QKV projection chunks at sequence boundaries and offsets its V^T scatter): same table, same scores."""
    import pandas as pd
    seq = str(golden["seq"])
    muts = list(pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))["mutant"])
    path = os.path.join(golden_dir, name + ".pt")
    m = pesm.load_model_and_alphabet(path)[0]
    a = pesm.Assay(m, seq, muts)
    s0, t0 = a.run(want_table=True)
    gemm_option("gemm_max_rows", 300)                          # a few sequences per chunk (T = len(seq) + 2 tokens each)
    s1, t1 = a.run(want_table=True)
    assert np.array_equal(s0, s1) and np.array_equal(t0, t1, equal_nan=True)
    a.close()
    m.close()


@pytest.mark.parametrize("precision", ["f16x3", "fp32", "bf16"])
@pytest.mark.parametrize("name", ["esm1v_toy_1", "esm1b_toy_lnb", "esm2_toy", "esm2_toy_h128"])
def test_last_layer_kept_rows_bit_identical(lib, golden, golden_dir, name, precision, monkeypatch):
    """Masked-marginals and pseudo-ppl read ONE output row per forward (compute_fitness.py:503, :274-276); the last layer's
    row-local stages run on those rows only.  Same bits as the full evaluation (PGMI_KEEP_ROWS=0)."""
    seq = str(golden["seq"])
    _, _, toks = pesm.Alphabet().get_batch_converter()([("p", seq)])
    n = toks.shape[1]
    rng = np.random.default_rng(5)
    pos = rng.permutation(n)                                  # kept rows in no particular order
    out = {}
    for keep in ("1", "0"):
        monkeypatch.setenv("PGMI_KEEP_ROWS", keep)
        m = pesm.load_model_and_alphabet(os.path.join(golden_dir, name + ".pt"), precision=precision)[0]
        out[keep] = m.masked_logprobs(np.repeat(toks, n, axis=0), pos)
        a = pesm.Assay(m, seq, ["%s%d%s" % (seq[i], i + 1, "A" if seq[i] != "A" else "C") for i in range(0, len(seq), 3)], offset_idx=1)
        out[keep + "s"] = a.run()
        m.close()
    assert np.array_equal(out["1"], out["0"])
    assert np.array_equal(np.asarray(out["1s"]), np.asarray(out["0s"]))


def test_esm1b_too_long_raises(models):
    toks = np.full((1, 1030), 5, np.int64)
    with pytest.raises(pesm.PgmiError, match="above maximum"):
        models["esm1v_toy_1"].token_logprobs(toks)


def test_wildtype_mismatch_raises(models, golden):
    seq = str(golden["seq"])
    bad = ("A" if seq[2] != "A" else "C") + "3G"
    with pytest.raises(AssertionError, match="does not match"):
        pesm.Assay(models["esm1v_toy_1"], seq, [bad])


@pytest.mark.parametrize("name", ["esm1v_toy_1", "esm1b_toy_lnb", "esm2_toy"])
def test_f16x3_mode_meets_the_parity_bar(lib, golden, golden_dir, name):
    """The split-fp16 mode is parity-gated like fp32: same 1e-4 bar against the reference."""
    import pandas as pd
    m = pesm.load_model_and_alphabet(os.path.join(golden_dir, name + ".pt"), precision="f16x3")[0]
    seq = str(golden["seq"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    a = pesm.Assay(m, seq, list(df["mutant"]), all_positions=True)
    scores, table = a.run(want_table=True)
    assert np.abs(table - golden[f"{name}/mm_table"]).max() < TOL
    assert np.abs(scores - golden[f"cli/{name}"]).max() < TOL
    toks = golden[f"{name}/pad_tokens"]
    lp = m.token_logprobs(toks)
    valid = toks != 1
    assert np.abs(lp[valid] - golden[f"{name}/pad_logprobs"][valid]).max() < TOL
    m.close()


def test_bf16_mode_runs_and_error_is_reported(lib, golden, golden_dir):
    """bf16 is the throughput mode: not parity-gated; its error is measured and bounded loosely."""
    import pandas as pd
    m = pesm.load_model_and_alphabet(os.path.join(golden_dir, "esm1v_toy_1.pt"), precision="bf16")[0]
    seq = str(golden["seq"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    scores = pesm.Assay(m, seq, list(df["mutant"])).run()
    err = np.abs(scores - golden["cli/esm1v_toy_1"]).max()
    print("bf16 toy max|err| =", err)
    assert err < 0.25 and err > 1e-6
    m.close()


def test_assay_outliving_its_model_is_inert(lib, golden, golden_dir):
    m = pesm.load_model_and_alphabet(os.path.join(golden_dir, "esm1v_toy_1.pt"))[0]
    a = pesm.Assay(m, str(golden["seq"]), ["%s1A" % str(golden["seq"])[0]] if str(golden["seq"])[0] != "A" else ["A1C"])
    a.run()
    m.close()
    with pytest.raises(pesm.PgmiError, match="destroyed model"):
        a.run()
    a.close()          # must not touch the freed model


# ---- ESM2 members with head_dim < 64 (8M / 35M / 150M: 16 / 24 / 32), run zero-padded to 64 lanes ----
@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("name", ["esm2_toy_h16", "esm2_toy_h24", "esm2_toy_h32"])
def test_small_head_dims_vs_reference(lib, golden_dir, name, precision):
    import pandas as pd
    g = np.load(os.path.join(golden_dir, "golden_esm_small_heads.npz"))
    seq = str(np.load(os.path.join(golden_dir, "golden_esm.npz"))["seq"])
    m, _ = pesm.load_model_and_alphabet(os.path.join(golden_dir, name + ".pt"), precision=precision)
    assert m.precision == precision                        # embed_dim 96 (h24: three K tiles of 32) runs in f16x3 too since round 6
    _, _, toks = pesm.Alphabet().get_batch_converter()([("p", seq)])
    assert np.abs(m(toks)["logits"][0] - g[f"{name}/wt_logprobs"]).max() < TOL
    n = toks.shape[1]
    lp = m.masked_logprobs(np.repeat(toks, n, axis=0), np.arange(n))
    assert np.abs(lp - g[f"{name}/mm_table"]).max() < TOL
    pt = g[f"{name}/pad_tokens"]
    valid = pt != 1
    assert np.abs(m.token_logprobs(pt)[valid] - g[f"{name}/pad_logprobs"][valid]).max() < TOL
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    a = pesm.Assay(m, seq, list(df["mutant"]), offset_idx=1)
    assert np.abs(a.run() - g[f"cli/{name}"]).max() < TOL
    a.close()
    m.close()


# ---- ESM2 with head_dim 128 (ESM2-15B: 48 x 5120, 40 heads), run as two 64-lane slot groups per head ----
@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_esm2_35m_width_f16x3_vs_reference(lib, golden_dir, tmp_path, precision):
    """ESM2-35M's width -- embed_dim 480 (15 K tiles of 32: not a multiple of 64, an ODD tile count for the ping-pong GEMM's two
    buffers), 20 heads of 24 (zero-padded to 64 slots), ffn 1920 -- in the parity-gated default mode against the UNMODIFIED
    reference's outputs (tests/golden/make_golden_esm2_35m_width.py; the checkpoint is rebuilt from its seed and must hash to the
    blob the reference ran on): wild-type log-probs, the whole masked-marginals table, a padded batch, the CLI's score column; flat
    1e-4.  Rounds 2-5 sent this registry row (/root/reference/config.json:18, esm/pretrained.py:355-360) to the 3x slower fp32 mode."""
    import hashlib
    import pandas as pd
    g = np.load(os.path.join(golden_dir, "golden_esm2_35m_width.npz"))
    seq = str(np.load(os.path.join(golden_dir, "golden_esm.npz"))["seq"])
    cfg = dict(synthetic.ESM2_35M, layers=3)
    blob = synthetic.random_weights(cfg, seed=35, embed_std=0.15)
    assert hashlib.sha256(blob.tobytes()).digest() == g["weights_sha256"].tobytes()
    path = synthetic.save_fair_esm_checkpoint(str(tmp_path / "esm2_toy_35m_width.pt"), cfg, blob)
    m, _ = pesm.load_model_and_alphabet(path, precision=precision)
    assert m.precision == precision
    _, _, toks = pesm.Alphabet().get_batch_converter()([("p", seq)])
    e_wt = float(np.abs(m(toks)["logits"][0] - g["wt_logprobs"]).max())
    n = toks.shape[1]
    e_mm = float(np.abs(m.masked_logprobs(np.repeat(toks, n, axis=0), np.arange(n)) - g["mm_table"]).max())
    pt = g["pad_tokens"]
    valid = pt != 1
    e_pad = float(np.abs(m.token_logprobs(pt)[valid] - g["pad_logprobs"][valid]).max())
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    a = pesm.Assay(m, seq, list(df["mutant"]), offset_idx=1)
    e_cli = float(np.abs(a.run() - g["cli"]).max())
    a.close()
    m.close()
    print(f"[{precision}] ESM2-35M width (480 x 20 heads of 24, 3 layers) vs the reference: wt {e_wt:.2e}, table {e_mm:.2e}, padded {e_pad:.2e}, CLI scores {e_cli:.2e}")
    assert max(e_wt, e_mm, e_pad, e_cli) < TOL


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_head_dim_128_vs_reference(lib, golden_dir, precision):
    """Reference-generated goldens (tests/golden/make_golden_h128.py: unmodified reference model and CLI on a 2-layer
    D=256, 2-head checkpoint): unmasked table, masked-marginals table, a padded batch and the CLI's score column -- in
    both parity-gated modes (fp32: the second mode an ESM2-15B user is sent to by PGMI_EOVERFLOW; attention_f32.hip DH = 128)."""
    import pandas as pd
    g = np.load(os.path.join(golden_dir, "golden_esm_h128.npz"))
    seq = str(np.load(os.path.join(golden_dir, "golden_esm.npz"))["seq"])
    m, _ = pesm.load_model_and_alphabet(os.path.join(golden_dir, "esm2_toy_h128.pt"), precision=precision)
    assert m.cfg["embed_dim"] // m.cfg["heads"] == 128 and m.precision == precision
    _, _, toks = pesm.Alphabet().get_batch_converter()([("p", seq)])
    assert np.abs(m(toks)["logits"][0] - g["wt_logprobs"]).max() < TOL
    n = toks.shape[1]
    lp = m.masked_logprobs(np.repeat(toks, n, axis=0), np.arange(n))
    assert np.abs(lp - g["mm_table"]).max() < TOL
    pt = g["pad_tokens"]
    valid = pt != 1
    assert np.abs(m.token_logprobs(pt)[valid] - g["pad_logprobs"][valid]).max() < TOL
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    a = pesm.Assay(m, seq, list(df["mutant"]), offset_idx=1)
    assert np.abs(a.run() - g["cli"]).max() < TOL
    a.close()
    m.close()


def test_esm2_15b_width_vs_oracle(lib):
    """ESM2-15B's own layer shape (5120 wide, 40 heads of 128, FFN 20480), 2 of its 48 layers, a 150-residue protein
    (T = 152: 5 key tiles, the last one partial), realistic-range weights: table rows and scores vs the fp32 oracle."""
    from oracle import esm_oracle as eo
    from proteingym_amd import synthetic
    cfg = dict(synthetic.ESM2_15B, layers=2)
    # tied embedding / LM-head rows N(0, 0.075^2): the logit spread of a random model grows with sqrt(width), real checkpoints'
    # does not; 0.15 at 1280 wide = 0.075 at 5120 wide keeps the log-prob range where real ESM2 models have it (~20).  With
    # 0.15 the range is 36 and the error 1.2e-4 (f16x3 operands carry 22 bits against fp32's 24: profiles/r2/README.md)
    blob = synthetic.random_weights(cfg, seed=15, embed_std=0.075)
    seq, muts, _ = synthetic.random_assay(seed=8, L=150, n_single=40, n_multi=10)
    m = pesm.EsmModel(cfg, blob, device=0)
    a = pesm.Assay(m, seq, muts)
    scores, table = a.run(want_table=True)
    positions = sorted(int(p) for p in a.positions)[:24]
    import torch
    import frozen

    def compute():
        tabs = {}
        for tag, dt in (("t32", torch.float32), ("t64", torch.float64)):
            ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), dtype=dt, **cfg)
            tabs[tag] = eo.masked_marginals_table(ocfg, W, seq, positions=positions, batch=8)
        return tabs
    fz = frozen.cached("esm2_15b_width_2_layers", ["ESM2_15B", 2, 15, 0.075, seq, positions, blob], compute)     # (tests/frozen.py)
    t32, t64 = fz["t32"], fz["t64"]
    noise = float(np.abs(t32[positions] - t64[positions]).max())          # the reference arithmetic's own distance to exact
    err64 = float(np.abs(table[positions] - t64[positions]).max())
    err32 = float(np.abs(table[positions] - t32[positions]).max())
    rng_lp = float(t64[positions][:, 4:24].max() - t64[positions][:, 4:24].min())
    print(f"ESM2-15B width (2 layers): HIP vs fp64 {err64:.2e}, HIP vs CPU fp32 {err32:.2e}, CPU fp32 vs fp64 {noise:.2e}, log-prob range {rng_lp:.1f}")
    # at K = 5120 / 20480 the CPU fp32 path itself is ~6e-5 away from exact arithmetic: the flat bar is held against the exact
    # result, and the distance to the fp32 oracle may exceed it by no more than that oracle's own error
    assert err64 < TOL
    assert err32 < TOL + noise
    a.close()
    m.close()


def test_real_small_esm2_shapes_load_and_match_oracle(lib):
    """ESM2 8M (6 x 320, 20 heads of 16) and 35M (12 x 480, 20 heads of 24) at their real shapes with
    synthetic weights against the oracle (no golden at this size; the oracle is pinned on the toys)."""
    from oracle import esm_oracle as eo
    from proteingym_amd import synthetic
    for layers, D in ((6, 320), (3, 480)):
        cfg = dict(synthetic.ESM2_650M, layers=layers, embed_dim=D, heads=20, ffn_dim=4 * D)
        blob = synthetic.random_weights(cfg, seed=D)
        seq, muts, _ = synthetic.random_assay(seed=3, L=50, n_single=40, n_multi=10)
        m = pesm.EsmModel(cfg, blob, device=0)
        scores = pesm.Assay(m, seq, muts).run()
        ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), **cfg)
        table = eo.masked_marginals_table(ocfg, W, seq, batch=8)
        ref = np.array([eo.label_row(mu, seq, table, 1) for mu in muts])
        assert np.abs(scores - ref).max() < TOL
        m.close()


@pytest.mark.parametrize("arch", ["ESM1V_650M", "ESM2_650M"])
def test_token_logprobs_do_not_depend_on_the_batch_or_on_the_gemm_item_kind(lib, gemm_option, arch):
    """The first sequences of a batch of 4 / 60 / 230 (4 layers at the 650M width: 1280 x 20 heads x 5120, T = 72): the small launches
    run every GEMM tile as two half-height items, the larger ones as full-height items (+ a half-height tail), the option gemm_half_tail = 0
    forces full-height items everywhere -- the same bits in all of them, through the fused QKV (rotary for ESM2), the split-plane
    (FC1 + GELU) and the fp32 + residual epilogues."""
    cfg = dict(getattr(synthetic, arch), layers=4)
    m = pesm.EsmModel(cfg, synthetic.random_weights(cfg, seed=3, embed_std=0.15), device=0)
    rng = np.random.default_rng(1)
    tok = rng.integers(4, 24, size=(230, 72)).astype(np.int32)
    tok[:, 0], tok[:, -1] = 0, 2
    gemm_option("gemm_half_tail", 1)
    base = m.token_logprobs(tok[:4])
    for half in (0, 1):
        gemm_option("gemm_half_tail", half)
        for B in (4, 60, 230):
            assert np.array_equal(m.token_logprobs(tok[:B])[:4], base), (half, B)
    m.close()


def test_spearman_vs_dms_score_identical_to_reference(models, golden, golden_dir):
    """the downstream metric (per-assay Spearman against DMS_score, 4 decimals --
    performance_DMS_benchmarks.py) computed from the HIP scores equals the one from the reference CLI's scores."""
    import pandas as pd
    from scipy.stats import spearmanr
    seq = str(golden["seq"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    for name in ("esm1v_toy_1", "esm2_toy", "esm1b_toy_lnb"):
        a = pesm.Assay(models[name], seq, list(df["mutant"]), offset_idx=1)
        mine = spearmanr(a.run(), df["DMS_score"]).correlation
        ref = spearmanr(golden[f"cli/{name}"], df["DMS_score"]).correlation
        a.close()
        assert round(mine, 4) == round(ref, 4)


@pytest.mark.parametrize("arch", ["esm1v", "esm2"])
def test_window_edges_around_1022_residues_vs_oracle(lib, arch):
    """The 1 024-token window is exactly full at 1 022 residues (compute_fitness.py:492-495, utils/scoring_utils.py:43-52): proteins
    of 1 021 ... 1 025 residues sit on both sides of that edge (no window; window == sequence; first protein that is cropped, by
    one and by three tokens).  Rows of the first / middle / last positions -- the three branches of get_optimal_window -- and a
    multi-mutant across them, against the oracle at a narrow width (the window logic, the learned-position offsets inside a
    cropped window and the rotary tables do not depend on the width)."""
    from oracle import esm_oracle as eo
    base = synthetic.ESM1V_650M if arch == "esm1v" else synthetic.ESM2_650M
    cfg = dict(base, layers=2, embed_dim=128, heads=4, ffn_dim=256)
    blob = synthetic.random_weights(cfg, seed=11, embed_std=0.3)
    ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), **cfg)
    model = pesm.EsmModel(cfg, blob, device=0, precision="f16x3")
    rng = np.random.default_rng(5)
    for L in (1021, 1022, 1023, 1025):
        seq = synthetic.random_sequence(rng, L)
        res = [1, 2, 511, 512, 513, L // 2, L - 512, L - 511, L - 1, L]          # 1-based residues = token positions
        aa = [a for a in synthetic.AA]
        subs = [f"{seq[p - 1]}{p}{next(a for a in aa if a != seq[p - 1])}" for p in res]
        muts = subs + [":".join(subs[::3])]
        assay = pesm.Assay(model, seq, muts)
        scores, table = assay.run(want_table=True)
        positions = [int(p) for p in assay.positions]
        assert positions == sorted(set(res))
        ref = eo.masked_marginals_table(ocfg, W, seq, positions=positions, batch=len(positions))
        err_t = float(np.abs(table[positions] - ref[positions]).max())
        ref_s = np.array([eo.label_row(m, seq, ref, 1) for m in muts])
        err_s = float(np.abs(scores - ref_s).max())
        print(f"[{arch}] L={L} (tokens {L + 2}): rows max|err| {err_t:.2e}, scores max|err| {err_s:.2e}")
        assert err_t < 1e-4 and err_s < 1e-4
        assay.close()
    model.close()


def test_esm2_beyond_1024_tokens_vs_oracle(lib):
    """ESM2 has no positional table, so the reference runs it at ANY length where it does not window: wt-marginals without
    --scoring-window overlapping (compute_fitness.py:476: the whole protein in one forward) and pseudo-ppl (:258-279: no window at
    all).  A 1 500-residue forward and the pseudo-ppl of a 1 030-residue sequence -- rotary angles and attention tiles beyond
    1 024 tokens -- against the oracle; the ESM-1b arch refuses the same input like the reference's positional embedding does."""
    import torch
    from oracle import esm_oracle as eo
    torch.set_num_threads(max(1, __import__("bench").usable_cores()))
    cfg = dict(synthetic.ESM2_650M, layers=2, embed_dim=128, heads=4, ffn_dim=256)
    blob = synthetic.random_weights(cfg, seed=13, embed_std=0.3)
    ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), **cfg)
    model = pesm.EsmModel(cfg, blob, device=0, precision="f16x3")
    rng = np.random.default_rng(8)
    seq = synthetic.random_sequence(rng, 1500)
    toks = eo.tokenize(seq)
    got = model.token_logprobs(toks[None])[0]
    with torch.no_grad():
        ref = torch.log_softmax(eo.forward_logits(ocfg, W, toks[None]), dim=-1)[0].numpy()
    err = float(np.abs(got - ref).max())
    print(f"ESM2 one forward of {len(toks)} tokens: max|err| {err:.2e}")
    assert err < TOL
    seq2 = synthetic.random_sequence(rng, 1030)
    lib2 = pesm.SequenceLibrary(model, [seq2])
    s, terms = lib2.score(want_terms=True)
    assert len(terms[0]) == len(seq2) - 2 and abs(float(s[0]) - float(np.sum(terms[0], dtype=np.float64))) < 1e-9
    # the terms at 20 positions (both ends, around token 1 024, beyond it), each from its own oracle forward with compute_pppl's
    # indexing (oracle/esm_oracle.py:293-305: token i masked, the letter sequence[i] looked up); the sum of all 1 028 terms carries
    # ~1e-3 of fp32 accumulation noise on either side (tests/test_gpu_parity_real_width.py), so the terms are what is compared
    t2 = eo.tokenize(seq2)
    worst = 0.0
    for i in [1, 2, 3, 511, 512, 1020, 1021, 1022, 1023, 1024, 1025, 1026, 1027, 1028] + [int(x) for x in rng.integers(4, 1028, 6)]:
        t = t2.copy()
        t[i] = eo.MASK
        with torch.no_grad():
            want = float(torch.log_softmax(eo.forward_logits(ocfg, W, t[None]), dim=-1)[0, i, eo.get_idx(seq2[i])])
        worst = max(worst, abs(float(terms[0][i - 1]) - want))
    print(f"ESM2 pseudo-ppl of a 1 030-residue sequence ({len(terms[0])} terms, 1 032 tokens): per-term max|err| {worst:.2e} at 20 positions")
    assert worst < TOL
    lib2.close()
    model.close()
    cfg1 = dict(synthetic.ESM1V_650M, layers=2, embed_dim=128, heads=4, ffn_dim=256)
    m1 = pesm.EsmModel(cfg1, synthetic.random_weights(cfg1, seed=13, embed_std=0.3), device=0, precision="f16x3")
    with pytest.raises(pesm.PgmiError, match="above maximum sequence length"):
        m1.token_logprobs(toks[None])
    m1.close()


@pytest.mark.parametrize("arch", ["ESM1V_650M", "ESM2_650M"])
def test_attention_launch_options_keep_the_bits_of_a_model(lib, arch):
    """Split-plane attention output (the model path) under the XCD-local block order and under the software-pipelined kernel (att_v3) against the
    4-wave kernel: 4 layers at the 650M width, T = 288 and a padded batch (key masks): the same token log-probs, bit for bit."""
    from proteingym_amd import _lib
    cfg = dict(getattr(synthetic, arch), layers=4)
    m = pesm.EsmModel(cfg, synthetic.random_weights(cfg, seed=3, embed_std=0.15), device=0)
    rng = np.random.default_rng(2)
    tok = rng.integers(4, 24, size=(40, 288)).astype(np.int32)
    tok[:, 0], tok[:, -1] = 0, 2
    tok[7, 200:] = 1
    tok[7, 199] = 2
    tok[21, 31:] = 1
    tok[21, 30] = 2
    try:
        _lib.check(lib.pgmi_set_option(b"att_xcd_local", 0))
        _lib.check(lib.pgmi_set_option(b"att_v3", 0))
        base = m.token_logprobs(tok)
        for name, value in ((b"att_xcd_local", 1), (b"att_xcd_local", -1), (b"att_v3", 1), (b"att_v3", -1)):
            _lib.check(lib.pgmi_set_option(name, value))
            got = m.token_logprobs(tok)
            keep = tok != 1
            assert np.array_equal(got[keep], base[keep]), (name, value)
    finally:
        lib.pgmi_set_option(b"att_xcd_local", -1)
        lib.pgmi_set_option(b"att_v3", -1)
        m.close()
