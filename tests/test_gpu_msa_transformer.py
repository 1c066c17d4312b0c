"""GPU parity of the MSA Transformer path (HIP, through the C ABI) against the reference's own outputs
(tests/golden/golden_msa_transformer.npz) and against the oracle at a wider shape."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from proteingym_amd import msa_transformer as pmsa

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "golden_msa_transformer.npz"))


@pytest.fixture(scope="module")
def model(lib, golden_dir):
    m, _ = pmsa.load_model_and_alphabet(os.path.join(golden_dir, "msa_toy.pt"), max_rows=32 * 1056)
    yield m
    m.close()


def test_logits_vs_reference(model, gold):
    ref = torch.log_softmax(torch.from_numpy(gold["logits"]), -1).numpy()
    lp = model.token_logprobs(gold["logits_tokens"])
    assert np.abs(lp - ref).max() < TOL
    out = model(gold["logits_tokens"][None])["logits"]
    assert out.shape == (1,) + ref.shape


def test_masked_marginals_table_vs_reference(model, gold):
    tok = gold["sampled/seed1"]
    rows = model.masked_logprobs(tok, np.arange(tok.shape[1]), seq_len=60)
    assert np.abs(rows - gold["mm_table/seed1"]).max() < TOL
    # a subset of positions gives the same rows; repeated calls are bit-identical (deterministic split-K order)
    sub = model.masked_logprobs(tok, [3, 17, 40], seq_len=60)
    assert np.array_equal(sub, rows[[3, 17, 40]])


def test_last_layer_kept_column_bit_identical(lib, gold, golden_dir, monkeypatch):
    """Masked-marginals reads token (row 0, masked column) only (compute_fitness.py:418-423): the last layer runs the row
    attention's out-projection and the column attention on that column's tokens, the feed-forward on that one token.
    Same bits as the full evaluation (PGMI_KEEP_ROWS=0), toy model and esm_msa1b's width at a ragged shape."""
    from proteingym_amd import synthetic
    tok = gold["sampled/seed1"]
    out = {}
    for keep in ("1", "0"):
        monkeypatch.setenv("PGMI_KEEP_ROWS", keep)
        m, _ = pmsa.load_model_and_alphabet(os.path.join(golden_dir, "msa_toy.pt"), max_rows=32 * 1056)
        out[keep] = m.masked_logprobs(tok, np.arange(tok.shape[1]), seq_len=60)
        m.close()
    assert np.array_equal(out["1"], out["0"])
    cfg = dict(arch=4, layers=2, embed_dim=768, heads=12, ffn_dim=3072, max_positions=1024, embed_positions_msa=True)
    blob = pmsa.pack_state_dict(cfg, synthetic.random_msa_transformer_arrays(cfg, seed=3))
    rng = np.random.default_rng(0)
    wide = rng.integers(4, 30, size=(70, 45)).astype(np.int64)
    wide[:, 0] = 0
    for keep in ("1", "0"):
        monkeypatch.setenv("PGMI_KEEP_ROWS", keep)
        m = pmsa.MsaTransformerModel(cfg, blob, max_rows=96 * 64)
        out[keep] = m.masked_logprobs(wide, [0, 1, 7, 31, 32, 44], seq_len=44)
        m.close()
    assert np.isfinite(out["1"]).all()
    assert np.array_equal(out["1"], out["0"])


def test_cli_matches_reference_columns(lib, gold, golden_dir, tmp_path):
    from proteingym_amd import compute_fitness as cf
    out = tmp_path / "o"
    os.makedirs(out)
    args = cf.create_parser().parse_args([
        "--model-location", os.path.join(golden_dir, "msa_toy.pt"), "--model_type", "MSA_transformer", "--dms_index", "0",
        "--dms_mapping", os.path.join(golden_dir, "TOY_MSA_MAPPING.csv"), "--dms-input", golden_dir, "--dms-output", str(out),
        "--scoring-strategy", "masked-marginals", "--scoring-window", "optimal", "--msa-path", golden_dir,
        "--msa-weights-folder", golden_dir, "--msa-samples", "12", "--seeds", "1", "2"])
    cf.main(args)
    df = pd.read_csv(out / "TOY_MSA_DMS.csv")
    assert list(df.columns) == list(gold["cli/columns"])
    for c in ("msa_toy_seed1", "msa_toy_seed2", "msa_toy_ensemble"):
        assert np.abs(df[c].to_numpy() - gold[f"cli/{c}"]).max() < TOL


def test_cli_pseudo_ppl_branch_matches_the_reference(lib, golden_dir, tmp_path):
    """--scoring-strategy pseudo-ppl with the MSA Transformer (compute_fitness.py:258-279, mode == "MSA_Transformer"; driver
    :403-417) against the reference CLI's own output (tests/golden/make_golden_msa_pppl.py): the mutated sequence in front of
    64 sampled rows, one alignment-wide forward per residue with alignment row i masked (the reference's indexing), sums of 58
    terms held to the flat 1e-4 bar; too few rows end in the reference's IndexError."""
    from proteingym_amd import compute_fitness as cf
    g = np.load(os.path.join(golden_dir, "golden_msa_pppl.npz"))
    src = pd.read_csv(os.path.join(golden_dir, "TOY_MSA_DMS.csv"))
    rows = src[src["mutant"].isin(list(g["mutants"]))].iloc[: len(g["mutants"])]
    assert list(rows["mutant"]) == list(g["mutants"])
    rows.to_csv(tmp_path / "TOY_MSA_PPPL.csv", index=False)
    mp = pd.read_csv(os.path.join(golden_dir, "TOY_MSA_MAPPING.csv"))
    mp["DMS_id"], mp["DMS_filename"] = "TOY_MSA_PPPL", "TOY_MSA_PPPL.csv"
    mp.to_csv(tmp_path / "map.csv", index=False)

    def run(samples, out):
        cf.main(cf.create_parser().parse_args([
            "--model-location", os.path.join(golden_dir, "msa_toy.pt"), "--model_type", "MSA_transformer", "--dms_index", "0",
            "--dms_mapping", str(tmp_path / "map.csv"), "--dms-input", str(tmp_path), "--dms-output", str(tmp_path / out),
            "--scoring-strategy", "pseudo-ppl", "--msa-path", golden_dir, "--msa-weights-folder", golden_dir,
            "--msa-samples", str(samples), "--seeds", "1", "2"]))
    run(64, "o")
    df = pd.read_csv(tmp_path / "o" / "TOY_MSA_PPPL.csv")
    assert list(df.columns) == list(g["cli/columns"])
    for c in ("msa_toy_seed1", "msa_toy_seed2", "msa_toy_ensemble"):
        err = np.abs(df[c].to_numpy() - g[f"cli/{c}"]).max()
        print(f"MSA Transformer pseudo-ppl {c}: max|err| {err:.2e} on sums of ~{g[f'cli/{c}'].mean():.0f}")
        assert err < TOL
    with pytest.raises(IndexError, match="out of bounds for dimension 1 with size 13"):
        run(12, "o2")


def test_long_alignment_optimal_window(lib, gold, golden_dir, tmp_path):
    from proteingym_amd import compute_fitness as cf
    seq_long = str(np.load(os.path.join(golden_dir, "golden_esm.npz"))["seq_long"])
    out = tmp_path / "o"
    os.makedirs(out)
    args = cf.create_parser().parse_args([
        "--model-location", os.path.join(golden_dir, "msa_toy.pt"), "--model_type", "MSA_transformer",
        "--dms-input", os.path.join(golden_dir, "TOY_MSA_LONG_DMS.csv"), "--dms-output", str(out), "--target_seq", seq_long,
        "--scoring-strategy", "masked-marginals", "--scoring-window", "optimal", "--msa-path", os.path.join(golden_dir, "TOY_MSA_LONG.a2m"),
        "--msa-weights-folder", golden_dir, "--weight_file_name", "TOY_MSA_LONG_weights.npy", "--msa-samples", "6", "--seeds", "1"])
    cf.main(args)
    df = pd.read_csv(out / "TOY_MSA_LONG_DMS.csv")
    assert np.abs(df["msa_toy_seed1"].to_numpy() - gold["cli_long/msa_toy_seed1"]).max() < TOL


def test_wider_shape_vs_oracle(lib):
    """esm_msa1b's width (D=768, 12 heads, F=3072), 2 layers, 70 rows x 45 columns: ragged against every tile
    size (rows not a multiple of 32, columns not a multiple of 32/128, split-K over the rows)."""
    from oracle import msa_transformer_oracle as mo
    from proteingym_amd import synthetic
    cfg = dict(arch=4, layers=2, embed_dim=768, heads=12, ffn_dim=3072, max_positions=1024, embed_positions_msa=True)
    arrays = synthetic.random_msa_transformer_arrays(cfg, seed=3)
    blob = pmsa.pack_state_dict(cfg, arrays)
    rng = np.random.default_rng(0)
    tok = rng.integers(4, 30, size=(70, 45)).astype(np.int64)
    tok[:, 0] = 0
    m = pmsa.MsaTransformerModel(cfg, blob, max_rows=96 * 64)
    ocfg, W = mo.from_arrays(arrays=arrays, **cfg)
    with torch.no_grad():
        ref = torch.log_softmax(mo.forward_logits(ocfg, W, tok), -1).numpy()
    lp = m.token_logprobs(tok)
    assert np.abs(lp - ref).max() < TOL
    m.close()


def _real_shape_fixture(golden_dir):
    import importlib.util
    path = os.path.join(golden_dir, "golden_msa_real_shape.npz")
    if not os.path.exists(path):
        pytest.skip("golden_msa_real_shape.npz not generated (tests/golden/make_golden_msa_real_shape.py)")
    spec = importlib.util.spec_from_file_location("make_golden_msa_real_shape", os.path.join(golden_dir, "make_golden_msa_real_shape.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)                       # token_grid / seeds: the fixture's inputs are regenerated, only oracle outputs are stored
    return np.load(path), gen


def test_real_shape_400_rows_287_columns_vs_oracle(lib, golden_dir):
    """esm_msa1b at its REAL shape -- 12 layers x 768 x 12 heads, 400 sampled rows x 287 columns (a BLAT-sized alignment, forwarded
    whole like the reference does: axial_attention.py:81-110 only chunks beyond max_tokens_per_msa at inference) -- against the CPU
    oracle's frozen rows: all 287 log-prob rows of the query from the unmasked forward and three masked-marginals rows
    (compute_fitness.py:380-394), flat 1e-4.  The fixture also holds the oracle in fp64: the reference's own fp32 distance."""
    from proteingym_amd import synthetic
    g, gen = _real_shape_fixture(golden_dir)
    cfg = dict(synthetic.MSA_1B)
    arrays = synthetic.random_msa_transformer_arrays(cfg, seed=gen.SEED_W)
    tok = gen.token_grid(287, 1)
    m = pmsa.MsaTransformerModel(cfg, pmsa.pack_state_dict(cfg, arrays), max_rows=416 * 288)
    lp = m.token_logprobs(tok)
    err_all = float(np.abs(lp[0] - g["c287/row0_logprobs"]).max())
    err64 = float(np.abs(lp[0] - g["c287/row0_logprobs_fp64"]).max())
    pos = [int(p) for p in g["c287/positions"]]
    rows = m.masked_logprobs(tok, np.array(pos), seq_len=286)
    err_mm = float(np.abs(rows - g["c287/mm_rows"]).max())
    print(f"MSA Transformer 12 x 768 x 12, 400 x 287: query-row log-probs max|err| {err_all:.2e} vs the fp32 oracle ({err64:.2e} vs fp64; the "
          f"oracle's own fp32 vs fp64: {float(g['c287/fp32_vs_fp64'][0]):.2e}); masked-marginals rows {err_mm:.2e}")
    m.close()
    assert err_all < TOL and err_mm < TOL


def test_real_shape_1024_column_window_vs_oracle(lib, golden_dir):
    """The same model on a 400 x 1100 alignment: position 700's optimal window (utils/scoring_utils.py:43-52) is a 400 x 1024 forward --
    the tied row attention's 1024 x 1024 score matrix per head, the operand planes at 1.26 GB -- against the oracle's frozen row."""
    from proteingym_amd import synthetic
    g, gen = _real_shape_fixture(golden_dir)
    cfg = dict(synthetic.MSA_1B)
    arrays = synthetic.random_msa_transformer_arrays(cfg, seed=gen.SEED_W)
    tok = gen.token_grid(1101, 2)
    m = pmsa.MsaTransformerModel(cfg, pmsa.pack_state_dict(cfg, arrays), max_rows=416 * 1024)
    i = int(g["c1100/position"][0])
    row = m.masked_logprobs(tok, np.array([i]), seq_len=1100)
    err = float(np.abs(row - g["c1100/mm_row"]).max())
    print(f"MSA Transformer 12 x 768 x 12, 400 x 1024-column window (position {i} of 1100): masked row max|err| {err:.2e}")
    m.close()
    assert err < TOL


def test_errors(model, gold):
    from proteingym_amd import _lib
    tok = gold["sampled/seed1"].copy()
    tok[2, 5] = 1
    with pytest.raises(_lib.PgmiError, match="equal length"):
        model.token_logprobs(tok)
    big = np.zeros((40, 1000), np.int64) + 5
    with pytest.raises(_lib.PgmiError, match="workspace"):
        model.token_logprobs(big)
