"""The oracle (oracle/esm_oracle.py) against the reference's own outputs.

* test_oracle_reproduces_golden*: against tests/golden/golden_esm.npz, frozen from the unmodified
  reference by tests/golden/make_golden.py -- runs everywhere (also on the GPU box).
* test_oracle_vs_live_reference: runs the reference itself (only where /root/reference exists).
"""
import ctypes as C
import os

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import esm_oracle as eo, ref_harness as rh

NAMES = ["esm1v_toy_1", "esm1v_toy_2", "esm1b_toy_lnb", "esm2_toy"]
TOL = 2e-5      # fp32 summation-order noise between two CPU implementations of the same formulae


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden_tables(golden, golden_dir, name):
    cfg, W = eo.load_checkpoint(os.path.join(golden_dir, name + ".pt"))
    seq = str(golden["seq"])
    with torch.no_grad():
        wt = torch.log_softmax(eo.forward_logits(cfg, W, eo.tokenize(seq)[None]), -1)[0].numpy()
    assert np.abs(wt - golden[f"{name}/wt_logprobs"]).max() < TOL
    mm = eo.masked_marginals_table(cfg, W, seq, batch=16)
    assert np.abs(mm - golden[f"{name}/mm_table"]).max() < TOL
    pt = golden[f"{name}/pad_tokens"]
    with torch.no_grad():
        lp = torch.log_softmax(eo.forward_logits(cfg, W, pt), -1).numpy()
    valid = pt != eo.PAD
    assert np.abs(lp[valid] - golden[f"{name}/pad_logprobs"][valid]).max() < TOL


def test_oracle_reproduces_golden_cli_scores(golden, golden_dir):
    seq = str(golden["seq"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    ck = [os.path.join(golden_dir, n + ".pt") for n in ("esm1v_toy_1", "esm1v_toy_2")]
    cols = eo.score_dms(ck, seq, list(df["mutant"]), model_type=["ESM1v"])
    for c in ("esm1v_toy_1", "esm1v_toy_2", "Ensemble_ESM1v"):
        assert np.abs(cols[c] - golden[f"cli/{c}"]).max() < TOL
    for n in ("esm2_toy", "esm1b_toy_lnb"):
        cols = eo.score_dms([os.path.join(golden_dir, n + ".pt")], seq, list(df["mutant"]), model_type=["X"])
        assert np.abs(cols[n] - golden[f"cli/{n}"]).max() < TOL
        assert "Ensemble_ESM1v" not in cols


def test_oracle_reproduces_golden_long_and_other_strategies(golden, golden_dir):
    seq, seq_long = str(golden["seq"]), str(golden["seq_long"])
    dfl = pd.read_csv(os.path.join(golden_dir, "TOY_LONG_DMS.csv"))
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    # optimal 1024-window for a 1102-token protein: only positions some mutant touches are needed
    cfg, W = eo.load_checkpoint(os.path.join(golden_dir, "esm2_toy.pt"))
    muts = list(dfl["mutant"])[:12]
    pos = sorted({1 + int(s[1:-1]) - 1 for m in muts for s in m.split(":")})
    table = eo.masked_marginals_table(cfg, W, seq_long, positions=pos)
    got = np.array([eo.label_row(m, seq_long, table, 1) for m in muts])
    assert np.abs(got - golden["cli_long/esm2_toy"][:12]).max() < TOL
    # wt-marginals, short and long/overlapping
    c = eo.score_dms([os.path.join(golden_dir, "esm1b_toy_lnb.pt")], seq, list(df["mutant"]),
                     strategy="wt-marginals", model_type=["ESM1b"])
    assert np.abs(c["esm1b_toy_lnb"] - golden["cli_wt/esm1b_toy_lnb"]).max() < TOL
    c = eo.score_dms([os.path.join(golden_dir, "esm1v_toy_1.pt")], seq_long, list(dfl["mutant"]),
                     strategy="wt-marginals", model_type=["ESM1b"], scoring_window="overlapping")
    assert np.abs(c["esm1v_toy_1"] - golden["cli_wt_long/esm1v_toy_1"]).max() < TOL
    # pseudo-ppl with the reference's off-by-one
    c = eo.score_dms([os.path.join(golden_dir, "esm2_toy.pt")], seq, list(df["mutant"])[:6],
                     strategy="pseudo-ppl", model_type=["ESM2"])
    assert np.abs(c["esm2_toy"] - golden["cli_pppl/esm2_toy"]).max() < 2e-4   # sum of ~68 terms


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_oracle_vs_live_reference(tmp_path):
    p = rh.make_esm1v_checkpoint(str(tmp_path / "esm1v_live.pt"), 2, 64, 128, 1, seed=9, embed_std=0.3)
    model, alphabet = rh.reference_model(p)
    cfg, W = eo.load_checkpoint(p)
    seq = "MKTAYIAKQRQISFVKSHFSRQLEERLGLIEVQAPILSRVGDGTQDNLSGAEKAVQ"
    _, _, bt = alphabet.get_batch_converter()([("x", seq)])
    assert np.array_equal(bt[0].numpy(), eo.tokenize(seq))
    t = eo.tokenize(seq).copy()
    t[7] = eo.MASK
    with torch.no_grad():
        ref = model(torch.tensor(t)[None])["logits"]
        mine = eo.forward_logits(cfg, W, t[None])
    assert float((ref - mine).abs().max()) < TOL
    cf = rh.load_reference()
    for i in (0, 10, 511, 512, 513, 600, 1500, 1988, 1989, 2500):
        for n in (100, 1024, 1025, 2000, 2501):
            if i < n:
                assert cf.get_optimal_window(i, n, 1024) == eo.get_optimal_window(i, n, 1024)


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_tokenizer_vs_live_reference():
    """esm/data.py:178-254 on odd inputs: whitespace dropped, literal special tokens recognised, a character outside the
    vocabulary is a KeyError (the product's Alphabet.encode / pack_sequences and the oracle's tokenize follow it)."""
    from proteingym_amd import esm as pesm
    rh.load_reference()
    from esm import data as ref_data
    ref = ref_data.Alphabet.from_architecture("ESM-1b")
    mine = pesm.Alphabet()
    for text in ["ACD", "A C D", "AC<mask>D", "A<mask>", "<cls>AC<eos>", "XBZ-.UO", "", "AC\nD", " A", "acd", "AJC", "A*C",
                 "ajJ K", "A<foo>C", "A<"]:
        try:
            want = ref.encode(text)
        except KeyError as e:
            with pytest.raises(KeyError) as got:
                mine.encode(text)
            assert got.value.args == e.args, text
            continue
        assert mine.encode(text) == want, text
        assert mine.tokenize(text) == ref.tokenize(text), text
        if all(len(t) == 1 for t in ref.tokenize(text)) and " " not in text and "\n" not in text:
            assert eo.tokenize(text).tolist() == [0] + want + [2]


# ---- Tranception -----------------------------------------------------------------------------------
def test_tranception_oracle_reproduces_golden(golden_dir):
    from oracle import tranception_oracle as to
    g = np.load(os.path.join(golden_dir, "golden_tranception.npz"))
    cfg, W = to.load_checkpoint(os.path.join(golden_dir, "Tranception_toy"))
    with torch.no_grad():
        lg = to.forward_logits(cfg, W, g["logits_ids"], g["logits_mask"]).numpy()
    m = g["logits_mask"].astype(bool)
    assert np.abs(lg - g["logits"])[m].max() < 5e-5
    seq = str(g["seq"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_DMS.csv"))
    r = pd.merge(df[["mutated_sequence"]], to.score_mutants(cfg, W, df, seq), on="mutated_sequence", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(r[c].to_numpy() - g[f"scores/{c}"]).max() < TOL
    ms, me = [int(v) for v in g["msa_start_end"]]
    prior = to.get_msa_prior(os.path.join(golden_dir, "TOY_MSA.a2m"), ms, me, len(seq))
    assert np.abs(prior - g["msa_prior"]).max() == 0.0
    retr = dict(log_prior=torch.log(torch.tensor(prior).float()).numpy(), MSA_start=ms, MSA_end=me, weight=0.6)
    r = pd.merge(df[["mutated_sequence"]], to.score_mutants(cfg, W, df, seq, retrieval=retr), on="mutated_sequence", how="left")
    assert np.abs(r["avg_score"].to_numpy() - g["scores_retrieval/avg_score"]).max() < TOL
    # slopes (model_pytorch.py:50-71): Tranception-L has 20 heads -> grouped over 5, not a power of two
    s20 = to.get_slopes(20, mode="grouped_alibi")
    assert len(s20) == 20 and s20[:5] == s20[5:10] and abs(s20[0] - 2 ** -2) < 1e-12


def test_tranception_host_slices_match_oracle(golden_dir):
    """Product host logic (proteingym_amd/tranception.py) against the oracle: windows, slices, prior."""
    from oracle import tranception_oracle as to
    from proteingym_amd import tranception as ptr
    g = np.load(os.path.join(golden_dir, "golden_tranception.npz"))
    seql = str(g["seq_long"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_LONG_DMS.csv"))[["mutated_sequence", "mutant"]]
    a = ptr.get_sequence_slices(df.copy(), seql, 1022)
    b = to.get_sequence_slices(df.copy(), seql, 1022).reset_index(drop=True)
    assert a[["mutated_sequence", "sliced_mutated_sequence", "window_start", "window_end"]].equals(
        b[["mutated_sequence", "sliced_mutated_sequence", "window_start", "window_end"]])
    ms, me = [int(v) for v in g["msa_start_end"]]
    p = ptr.get_msa_prior(os.path.join(golden_dir, "TOY_MSA.a2m"), None, ms, me, 70)
    assert np.array_equal(p, g["msa_prior"])
    cfg, blob = ptr.load_checkpoint(os.path.join(golden_dir, "Tranception_toy"))
    assert cfg["layers"] == 2 and cfg["embed_dim"] == 256 and cfg["heads"] == 4 and cfg["vocab"] == 25
    from proteingym_amd import _lib
    import ctypes as C
    c = _lib.Config(abi_version=_lib.ABI_VERSION, arch=3, layers=2, embed_dim=256, heads=4, ffn_dim=1024, vocab=25,
                    max_positions=1024, token_dropout=0, emb_layer_norm_before=0, precision=2, max_rows=0, ln_eps=1e-5)
    _lib.load()
    assert _lib.load().pgmi_weight_count(C.byref(c)) == blob.size


def test_eve_sequence_weights_oracle_and_host_match_reference(golden_dir, tmp_path):
    """MSA_processing (msa_utils.py:194-368): the oracle's plain-loop restatement reproduces the weights the reference computed
    (TOY_MSA_GAPPY_weights.npy); the product's mirror keeps the same set / order of sequences and builds the same weighted prior."""
    import shutil
    from oracle import tranception_oracle as to
    from proteingym_amd import tranception as ptr
    g = np.load(os.path.join(golden_dir, "golden_msa_weights.npz"))
    a2m = os.path.join(golden_dir, "TOY_MSA_GAPPY.a2m")
    wfile = os.path.join(golden_dir, "TOY_MSA_GAPPY_weights.npy")
    assert np.array_equal(np.load(wfile), g["weights"])
    w = to.eve_sequence_weights(a2m)
    assert list(w.keys()) == list(g["names"])
    assert np.abs(np.array(list(w.values())) - g["weights"]).max() == 0.0
    # host: the product LOADS an existing weights file here; computing one is the HIP kernel's job (tests/test_gpu_msa_weights.py)
    mp = ptr.MSA_processing(MSA_location=a2m, use_weights=True, weights_location=wfile)
    assert list(mp.seq_name_to_weight.keys()) == list(g["names"])
    assert np.array_equal(mp.weights, g["weights"]) and abs(mp.Neff - float(g["Neff"])) < 1e-12
    ms, me = [int(v) for v in g["msa_start_end"]]
    assert np.array_equal(ptr.get_msa_prior(a2m, wfile, ms, me, 70), g["msa_prior"])
    assert np.abs(to.get_msa_prior(a2m, ms, me, 70, weights=w) - g["msa_prior"]).max() == 0.0
    # oracle retrieval scores with the weighted prior
    gt = np.load(os.path.join(golden_dir, "golden_tranception.npz"))
    seq = str(gt["seq"])
    cfg, W = to.load_checkpoint(os.path.join(golden_dir, "Tranception_toy"))
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_DMS.csv"))
    retr = dict(log_prior=torch.log(torch.tensor(g["msa_prior"]).float()).numpy(), MSA_start=ms, MSA_end=me, weight=0.6)
    r = pd.merge(df[["mutated_sequence"]], to.score_mutants(cfg, W, df, seq, retrieval=retr), on="mutated_sequence", how="left")
    assert np.abs(r["avg_score"].to_numpy() - g["scores_retrieval_weighted/avg_score"]).max() < TOL


# ---- alignment pair count / sequence weights (proteingym/utils/weights.py) --------------------------
MSA_CASES = ["small", "ragged", "wide", "thr_edge", "thr_07", "thr_1m02"]


@pytest.mark.parametrize("name", MSA_CASES)
def test_msa_cluster_oracle_reproduces_reference(golden_dir, name):
    from oracle import msa_weights_oracle as mo
    g = np.load(os.path.join(golden_dir, "golden_msa_cluster.npz"))
    m, thr = g[f"{name}/matrix"], float(g[f"{name}/threshold"])
    assert np.array_equal(mo.cluster_counts(m, thr, 20), g[f"{name}/counts"])
    assert np.array_equal(mo.cluster_counts_numpy(m, thr, 20), g[f"{name}/counts"])
    assert np.abs(mo.calc_weights(m, thr, 20) - g[f"{name}/weights"]).max() == 0.0


def test_msa_cluster_oracle_vs_live_reference():
    from oracle import msa_weights_oracle as mo, ref_harness as rh
    if not rh.reference_available():
        pytest.skip("reference checkout not present")
    w = rh.load_reference_weights()
    rng = np.random.default_rng(3)
    m = rng.integers(0, 21, size=(50, 23)).astype(np.int64)
    m[rng.integers(0, 50, size=20)] = m[0]                      # a cluster of duplicates
    ref = w.calc_num_cluster_members_nogaps_parallel(m, 0.8, 20)
    assert np.array_equal(mo.cluster_counts(m, 0.8, 20), ref.astype(np.int32))


@pytest.mark.parametrize("name", ["esm2_toy_h16", "esm2_toy_h24", "esm2_toy_h32"])
def test_oracle_reproduces_small_head_goldens(golden_dir, name):
    """ESM2 with head_dim 16 / 24 / 32 (rotary over the true head dim)."""
    g = np.load(os.path.join(golden_dir, "golden_esm_small_heads.npz"))
    seq = str(np.load(os.path.join(golden_dir, "golden_esm.npz"))["seq"])
    cfg, W = eo.load_checkpoint(os.path.join(golden_dir, name + ".pt"))
    table = eo.masked_marginals_table(cfg, W, seq, batch=16)
    assert np.abs(table - g[f"{name}/mm_table"]).max() < 2e-5
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    scores = np.array([eo.label_row(m, seq, table, 1) for m in df["mutant"]])
    assert np.abs(scores - g[f"cli/{name}"]).max() < 2e-5


def test_oracle_reproduces_the_esm2_35m_width_golden(golden_dir, tmp_path):
    """ESM2-35M's width (embed_dim 480 = 15 K tiles of 32, 20 heads of 24): the checkpoint is rebuilt from its seed (the fixture
    carries the sha256 of the weight blob the reference ran on), the oracle reproduces the reference's table and CLI column."""
    import hashlib
    from proteingym_amd import synthetic
    g = np.load(os.path.join(golden_dir, "golden_esm2_35m_width.npz"))
    seq = str(np.load(os.path.join(golden_dir, "golden_esm.npz"))["seq"])
    cfg = dict(synthetic.ESM2_35M, layers=3)
    blob = synthetic.random_weights(cfg, seed=35, embed_std=0.15)
    assert hashlib.sha256(blob.tobytes()).digest() == g["weights_sha256"].tobytes()
    path = synthetic.save_fair_esm_checkpoint(str(tmp_path / "esm2_toy_35m_width.pt"), cfg, blob)
    ocfg, W = eo.load_checkpoint(path)
    table = eo.masked_marginals_table(ocfg, W, seq, batch=16)
    assert np.abs(table - g["mm_table"]).max() < 2e-5
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    scores = np.array([eo.label_row(m, seq, table, 1) for m in df["mutant"]])
    assert np.abs(scores - g["cli"]).max() < 5e-5          # sums of up to five fp32-noise terms (the table rows hold 2e-5; the CLI ran batch 1)


# ---- MSA Transformer ---------------------------------------------------------------------------------
def test_msa_transformer_oracle_reproduces_golden(golden_dir):
    from oracle import msa_transformer_oracle as mo
    g = np.load(os.path.join(golden_dir, "golden_msa_transformer.npz"))
    cfg, W = mo.load_checkpoint(os.path.join(golden_dir, "msa_toy.pt"))
    with torch.no_grad():
        lg = mo.forward_logits(cfg, W, g["logits_tokens"]).numpy()
    assert np.abs(lg - g["logits"]).max() < 2e-5
    wts = np.load(os.path.join(golden_dir, "TOY_MSA_GAPPY_weights.npy"))
    pm = mo.ProcessedMSA(os.path.join(golden_dir, "TOY_MSA_GAPPY.a2m"), weights=wts)
    for seed in (1, 2):
        assert np.array_equal(mo.tokenize_msa(mo.sample_msa(pm, 12, seed)), g[f"sampled/seed{seed}"])
    table = mo.masked_marginals_table(cfg, W, g["sampled/seed1"], 60)
    assert np.abs(table - g["mm_table/seed1"]).max() < 2e-5
    seq = str(np.load(os.path.join(golden_dir, "golden_esm.npz"))["seq"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_MSA_DMS.csv"))
    cols = mo.score_dms(os.path.join(golden_dir, "msa_toy.pt"), os.path.join(golden_dir, "TOY_MSA_GAPPY.a2m"), wts,
                        seq[5:65], list(df["mutant"]), 6, [1, 2], 12)
    for k, c in (("seed1", "msa_toy_seed1"), ("seed2", "msa_toy_seed2"), ("ensemble", "msa_toy_ensemble")):
        assert np.abs(cols[k] - g[f"cli/{c}"]).max() < 2e-5


def test_msa_transformer_oracle_long_alignment_window(golden_dir):
    """1100 columns: every masked position is scored in its optimal 1024-column window (incl. the reference's
    end index running one past the token grid)."""
    import re
    from oracle import msa_transformer_oracle as mo
    g = np.load(os.path.join(golden_dir, "golden_msa_transformer.npz"))
    sl = str(np.load(os.path.join(golden_dir, "golden_esm.npz"))["seq_long"])
    cfg, W = mo.load_checkpoint(os.path.join(golden_dir, "msa_toy.pt"))
    dl = pd.read_csv(os.path.join(golden_dir, "TOY_MSA_LONG_DMS.csv"))
    pm = mo.ProcessedMSA(os.path.join(golden_dir, "TOY_MSA_LONG.a2m"), weights=np.load(os.path.join(golden_dir, "TOY_MSA_LONG_weights.npy")))
    tok = mo.tokenize_msa(mo.sample_msa(pm, 6, 1))
    pos = sorted({int(re.findall(r"\d+", m)[0]) for mm in dl["mutant"] for m in mm.split(":")})
    tab = mo.masked_marginals_table(cfg, W, tok, len(sl), positions=pos)
    sc = np.array([eo.label_row(m, sl, tab, 1) for m in dl["mutant"]])
    assert np.abs(sc - g["cli_long/msa_toy_seed1"]).max() < 2e-5


def test_msa_transformer_host_logic_matches_reference(golden_dir, tmp_path):
    """Product host code (proteingym_amd/msa_transformer.py): alignment pre-processing, weighted sampling with
    python's RNG, batch conversion, checkpoint key upgrade -> same token grids as the reference sampled."""
    from proteingym_amd import msa_transformer as pmsa, _lib
    g = np.load(os.path.join(golden_dir, "golden_msa_transformer.npz"))
    gw = np.load(os.path.join(golden_dir, "golden_msa_weights.npz"))
    mp = pmsa.MSA_processing(MSA_location=os.path.join(golden_dir, "TOY_MSA_GAPPY.a2m"), use_weights=True,
                             weights_location=os.path.join(golden_dir, "TOY_MSA_GAPPY_weights.npy"))
    assert list(mp.seq_name_to_weight.keys()) == list(gw["names"])
    conv = pmsa.MsaAlphabet().get_batch_converter()
    for seed in (1, 2):
        data = [pmsa.sample_msa(None, 12, "sequence-reweighting", seed, processed_msa=mp)]
        assert np.array_equal(conv(data)[2][0], g[f"sampled/seed{seed}"])
    first = pmsa.sample_msa(os.path.join(golden_dir, "TOY_MSA_GAPPY.a2m"), 3, "first_x_rows", 1)
    assert first[0][0] == "TARGET/6-65" and len(first) == 3
    cfg, sd = pmsa._upgrade_state_dict(os.path.join(golden_dir, "msa_toy.pt"))
    blob = pmsa.pack_state_dict(cfg, sd)
    c = _lib.Config(abi_version=_lib.ABI_VERSION, arch=4, layers=cfg["layers"], embed_dim=cfg["embed_dim"], heads=cfg["heads"],
                    ffn_dim=cfg["ffn_dim"], vocab=33, max_positions=cfg["max_positions"], token_dropout=0,
                    emb_layer_norm_before=1, precision=2, max_rows=0, ln_eps=0.0)
    import ctypes as C
    assert _lib.load().pgmi_weight_count(C.byref(c)) == blob.size
    with pytest.raises(RuntimeError, match="unaligned"):
        conv([("a", "MKV"), ("b", "MK")])


def test_tranception_oracle_indel_and_sliding_modes(golden_dir):
    """Indel scoring (variable-length sequences, WT row appended under 'mutant') and the 'sliding' window on a
    1100-residue protein: oracle vs the reference outputs frozen by make_golden_tranception_modes.py."""
    from oracle import tranception_oracle as to
    g = np.load(os.path.join(golden_dir, "golden_tranception_modes.npz"))
    gt = np.load(os.path.join(golden_dir, "golden_tranception.npz"))
    seq, seql = str(gt["seq"]), str(gt["seq_long"])
    cfg, W = to.load_checkpoint(os.path.join(golden_dir, "Tranception_toy"))
    indel = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_INDEL_DMS.csv"))
    r = to.score_mutants(cfg, W, indel, seq, indel_mode=True)
    assert sorted(r.columns) == sorted(g["indel/columns"])
    wt = r[r["mutated_sequence"].isna()]
    assert len(wt) == 1 and wt["mutant"].iloc[0] == seq and float(wt["avg_score"].iloc[0]) == 0.0
    rr = pd.merge(indel[["mutated_sequence"]].iloc[1:], r, on="mutated_sequence", how="left")
    dl = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_LONG_DMS.csv"))
    rs = pd.merge(dl[["mutated_sequence"]], to.score_mutants(cfg, W, dl, seql, scoring_window="sliding"), on="mutated_sequence", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(rr[c].to_numpy() - g[f"indel/{c}"]).max() < TOL
        assert np.abs(rs[c].to_numpy() - g[f"sliding/{c}"]).max() < TOL


def test_tranception_oracle_retrieval_on_long_protein(golden_dir):
    """Retrieval with windows that overlap the alignment span differently (1100 residues, alignment 301..900)."""
    from oracle import tranception_oracle as to
    g = np.load(os.path.join(golden_dir, "golden_tranception_long_retrieval.npz"))
    seql = str(np.load(os.path.join(golden_dir, "golden_tranception.npz"))["seq_long"])
    ms, me = [int(v) for v in g["msa_start_end"]]
    cfg, W = to.load_checkpoint(os.path.join(golden_dir, "Tranception_toy"))
    prior = to.get_msa_prior(os.path.join(golden_dir, "TOY_MSA_LONGSPAN.a2m"), ms, me, len(seql))
    retr = dict(log_prior=torch.log(torch.tensor(prior).float()).numpy(), MSA_start=ms, MSA_end=me, weight=0.6)
    dl = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_LONG_DMS.csv"))
    r = pd.merge(dl[["mutated_sequence"]], to.score_mutants(cfg, W, dl, seql, retrieval=retr), on="mutated_sequence", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(r[c].to_numpy() - g[f"scores/{c}"]).max() < TOL


class _OracleBackedTranception:
    """The product's host scoring logic (TranceptionModel.score_mutants / _scores: slicing, length normalisation,
    WT delta per window, sliding aggregation, mirror averaging, WT row) with the ORACLE standing in for the device
    call ``sequence_loglik`` -- checks the host half of the product on CPU against the reference goldens."""

    def __new__(cls, cfg, W, scoring_window="optimal", retrieval=None):
        from oracle import tranception_oracle as to
        from proteingym_amd import tranception as ptr
        obj = object.__new__(ptr.TranceptionModel)
        obj._h = None
        obj.n_ctx = cfg["n_ctx"]
        obj.scoring_window = scoring_window
        obj.retrieval = retrieval

        def sequence_loglik(seqs, window_start=None, window_end=None, reverse=False, **kw):
            return np.asarray(to.sequence_scores(cfg, W, list(seqs), list(window_start), list(window_end), reverse=reverse,
                                                 retrieval=retrieval), dtype=np.float32)
        obj.sequence_loglik = sequence_loglik
        return obj


def test_tranception_product_host_logic_on_cpu(golden_dir):
    from oracle import tranception_oracle as to
    g = np.load(os.path.join(golden_dir, "golden_tranception.npz"))
    gm = np.load(os.path.join(golden_dir, "golden_tranception_modes.npz"))
    gl = np.load(os.path.join(golden_dir, "golden_tranception_long_retrieval.npz"))
    seq, seql = str(g["seq"]), str(g["seq_long"])
    cfg, W = to.load_checkpoint(os.path.join(golden_dir, "Tranception_toy"))
    tol = 2e-5                                                        # oracle fp32 vs reference fp32, scores O(0.1..1)

    def check(model, df, target, gold, prefix, **kw):
        r = model.score_mutants(DMS_data=df, target_seq=target, scoring_mirror=True, **kw)
        r = pd.merge(df[["mutated_sequence"]] if "mutated_sequence" in df else
                     pd.DataFrame({"mutated_sequence": [to.get_mutated_sequence(target, m) for m in df["mutant"]]}),
                     r, on="mutated_sequence", how="left")
        for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
            assert np.abs(r[c].to_numpy() - gold[f"{prefix}/{c}"]).max() < tol, (prefix, c)

    dms = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_DMS.csv"))
    dml = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_LONG_DMS.csv"))
    check(_OracleBackedTranception(cfg, W), dms, seq, g, "scores")
    check(_OracleBackedTranception(cfg, W), dml, seql, g, "scores_long")
    check(_OracleBackedTranception(cfg, W, scoring_window="sliding"), dml, seql, gm, "sliding")
    indel = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_INDEL_DMS.csv"))
    r = _OracleBackedTranception(cfg, W).score_mutants(DMS_data=indel, target_seq=seq, scoring_mirror=True, indel_mode=True)
    rr = pd.merge(indel[["mutated_sequence"]].iloc[1:], r, on="mutated_sequence", how="left")
    assert np.abs(rr["avg_score"].to_numpy() - gm["indel/avg_score"]).max() < tol
    assert sorted(r.columns) == sorted(gm["indel/columns"])
    ms, me = [int(v) for v in gl["msa_start_end"]]
    prior = to.get_msa_prior(os.path.join(golden_dir, "TOY_MSA_LONGSPAN.a2m"), ms, me, len(seql))
    retr = dict(log_prior=torch.log(torch.tensor(prior).float()).numpy(), MSA_start=ms, MSA_end=me, weight=0.6)
    check(_OracleBackedTranception(cfg, W, retrieval=retr), dml, seql, gl, "scores")


class _OracleBackedEsm:
    """token_logprobs / masked_logprobs of the product's EsmModel served by the oracle: lets the CLI mirror's host
    logic (window blending of wt-marginals, pseudo-ppl batching, label_row) run on CPU against the reference goldens."""

    def __init__(self, path):
        self.cfg, self.W = eo.load_checkpoint(path)

    def token_logprobs(self, tokens):
        with torch.no_grad():
            return torch.log_softmax(eo.forward_logits(self.cfg, self.W, np.asarray(tokens)), -1).numpy()

    def masked_logprobs(self, tokens, mask_pos):
        t = np.array(tokens, copy=True)
        t[np.arange(len(t)), np.asarray(mask_pos)] = eo.MASK
        out = []
        with torch.no_grad():
            for b0 in range(0, len(t), 64):
                lp = torch.log_softmax(eo.forward_logits(self.cfg, self.W, t[b0:b0 + 64]), -1).numpy()
                out.append(lp[np.arange(len(lp)), np.asarray(mask_pos)[b0:b0 + 64]])
        return np.concatenate(out)


def test_esm_cli_host_logic_on_cpu(golden, golden_dir):
    """wt-marginals (short and the overlapping-window blend of a 1100-residue protein) through the
    product's host functions with oracle-backed forwards, against the reference CLI's score columns."""
    from proteingym_amd import compute_fitness as cf, esm as pesm
    alphabet = pesm.Alphabet()
    seq, seql = str(golden["seq"]), str(golden["seq_long"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    dfl = pd.read_csv(os.path.join(golden_dir, "TOY_LONG_DMS.csv"))
    m = _OracleBackedEsm(os.path.join(golden_dir, "esm1b_toy_lnb.pt"))
    table = cf.wt_marginals_table(m, alphabet, seq, "optimal")
    got = np.array([cf.label_row(x, seq, table, alphabet, 1) for x in df["mutant"]])
    assert np.abs(got - golden["cli_wt/esm1b_toy_lnb"]).max() < 2e-5
    m = _OracleBackedEsm(os.path.join(golden_dir, "esm1v_toy_1.pt"))
    table = cf.wt_marginals_table(m, alphabet, seql, "overlapping")
    got = np.array([cf.label_row(x, seql, table, alphabet, 1) for x in dfl["mutant"]])
    assert np.abs(got - golden["cli_wt_long/esm1v_toy_1"]).max() < 2e-5
    # pseudo-ppl: the (sequence, position) enumeration lives in the C library now (pgmi_pppl_*, GPU tests in
    # tests/test_gpu_pppl.py); the host side left is get_mutated_sequence (:252-257)
    muts = list(df["mutant"][:6])
    seqs = [cf.get_mutated_sequence(x, seq, 1) for x in muts]
    assert all(len(s) == len(seq) and sum(a != b for a, b in zip(s, seq)) == 1 for s in seqs)


def test_msa_transformer_cli_host_logic_on_cpu(golden_dir, tmp_path, monkeypatch):
    """The whole MSA Transformer branch of the CLI mirror (alignment pre-processing, per-seed sampling, masked cells,
    label_row, per-seed columns, ensemble, CSV layout, skip-existing-seed resume) with the oracle standing in for the
    device model, against the reference CLI's columns."""
    from oracle import msa_transformer_oracle as mo
    from proteingym_amd import compute_fitness as cf, msa_transformer as pmsa
    g = np.load(os.path.join(golden_dir, "golden_msa_transformer.npz"))
    calls = []

    class Fake:
        def __init__(self, path):
            self.cfg, self.W = mo.load_checkpoint(path)

        def masked_logprobs(self, tokens, positions, seq_len, window=1024):
            calls.append(len(positions))
            table = mo.masked_marginals_table(self.cfg, self.W, np.asarray(tokens, dtype=np.int64), seq_len, positions=list(positions))
            return table[list(positions)]

        def close(self):
            pass

    monkeypatch.setattr(pmsa, "load_model_and_alphabet", lambda loc, device=0, max_rows=0: (Fake(loc), pmsa.MsaAlphabet()))
    out = tmp_path / "o"
    argv = ["--model-location", os.path.join(golden_dir, "msa_toy.pt"), "--model_type", "MSA_transformer", "--dms_index", "0",
            "--dms_mapping", os.path.join(golden_dir, "TOY_MSA_MAPPING.csv"), "--dms-input", golden_dir, "--dms-output", str(out),
            "--scoring-strategy", "masked-marginals", "--scoring-window", "optimal", "--msa-path", golden_dir,
            "--msa-weights-folder", golden_dir, "--msa-samples", "12", "--seeds", "1", "2"]
    cf.main(cf.create_parser().parse_args(argv))
    df = pd.read_csv(out / "TOY_MSA_DMS.csv")
    assert list(df.columns) == list(g["cli/columns"])
    for c in ("msa_toy_seed1", "msa_toy_seed2", "msa_toy_ensemble"):
        assert np.abs(df[c].to_numpy() - g[f"cli/{c}"]).max() < 2e-5
    n_calls = len(calls)
    cf.main(cf.create_parser().parse_args(argv))                      # both seed columns exist: nothing is recomputed
    assert len(calls) == n_calls
    again = pd.read_csv(out / "TOY_MSA_DMS.csv")
    assert np.allclose(again["msa_toy_ensemble"], df["msa_toy_ensemble"], rtol=0, atol=1e-12)


def test_esm_cli_main_masked_marginals_on_cpu(golden, golden_dir, tmp_path, monkeypatch):
    """compute_fitness.main() for the ESM-1v ensemble (two checkpoints): argument resolution, column naming from the
    checkpoint stems, plain-mean ensemble and CSV layout, with the oracle standing in for the device (model + Assay)."""
    from proteingym_amd import compute_fitness as cf, esm as pesm

    class FakeModel(_OracleBackedEsm):
        def close(self):
            pass

    class FakeAssay:
        def __init__(self, model, sequence, mutants, offset_idx=1, alphabet=None, window=1024, all_positions=False, positions=None):
            self.args = (model, sequence, list(mutants), offset_idx)

        def run(self):
            model, sequence, mutants, offset = self.args
            pos = pesm.positions_read(mutants, sequence, offset)
            table = eo.masked_marginals_table(model.cfg, model.W, sequence, positions=[int(p) for p in pos], batch=16)
            return pesm.score_from_table(table, mutants, sequence, offset)

        def close(self):
            pass

    monkeypatch.setattr(pesm, "load_model_and_alphabet", lambda loc, device=0, precision="f16x3", max_rows=0: (FakeModel(loc), pesm.Alphabet()))
    monkeypatch.setattr(pesm, "Assay", FakeAssay)
    out = tmp_path / "o"
    cf.main(cf.create_parser().parse_args([
        "--model-location", os.path.join(golden_dir, "esm1v_toy_1.pt"), os.path.join(golden_dir, "esm1v_toy_2.pt"),
        "--model_type", "ESM1v", "--dms-input", os.path.join(golden_dir, "TOY_DMS.csv"), "--dms-output", str(out),
        "--target_seq", str(golden["seq"]), "--scoring-strategy", "masked-marginals", "--scoring-window", "optimal"]))
    df = pd.read_csv(out / "TOY_DMS.csv")
    assert list(df.columns) == list(golden["cli/columns"])
    for c in ("esm1v_toy_1", "esm1v_toy_2", "Ensemble_ESM1v"):
        assert np.abs(df[c].to_numpy() - golden[f"cli/{c}"]).max() < 2e-5


def test_oracle_head_dim_128_reproduces_reference(golden_dir, golden):
    """ESM2 with head_dim 128 (the ESM2-15B class): oracle vs the reference-generated goldens of make_golden_h128.py."""
    g = np.load(os.path.join(golden_dir, "golden_esm_h128.npz"))
    seq = str(golden["seq"])
    cfg, W = eo.load_checkpoint(os.path.join(golden_dir, "esm2_toy_h128.pt"))
    assert cfg["embed_dim"] // cfg["heads"] == 128
    with torch.no_grad():
        lp = torch.log_softmax(eo.forward_logits(cfg, W, eo.tokenize(seq)[None]), -1)[0].numpy()
    assert np.abs(lp - g["wt_logprobs"]).max() < 2e-5
    table = eo.masked_marginals_table(cfg, W, seq, batch=8)
    assert np.abs(table - g["mm_table"]).max() < 2e-5
    df = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    got = np.array([eo.label_row(m, seq, table, 1) for m in df["mutant"]])
    assert np.abs(got - g["cli"]).max() < 5e-5


@pytest.mark.parametrize("n_tok", [1025, 1537, 1538, 2048, 3000, 3427])
def test_wt_marginals_overlapping_windows_beyond_two(golden_dir, n_tok):
    """wt-marginals with overlapping windows on proteins that take the stepping loop and the central window
    (compute_fitness.py:433-475; tests/golden/make_golden_wt_overlapping.py ran the unmodified reference CLI): the oracle's
    restatement, the product's host blend over oracle-backed forwards, and the runner's window count, branch by branch."""
    from proteingym_amd import compute_fitness as cf, esm as pesm, run_benchmark as rb
    g = np.load(os.path.join(golden_dir, "golden_wt_overlapping.npz"))
    seq, muts, want = str(g[f"{n_tok}/seq"]), [str(m) for m in g[f"{n_tok}/mutants"]], g[f"{n_tok}/scores"]
    assert len(seq) + 2 == n_tok
    ck = os.path.join(golden_dir, "esm1v_toy_1.pt")
    got = eo.score_dms([ck], seq, muts, strategy="wt-marginals", model_type=["ESM1b"], scoring_window="overlapping")["esm1v_toy_1"]
    assert np.abs(got - want).max() < 2e-5

    class Counting(_OracleBackedEsm):
        windows = 0

        def token_logprobs(self, tokens):
            Counting.windows += len(tokens)
            return super().token_logprobs(tokens)

    table = cf.wt_marginals_table(Counting(ck), pesm.Alphabet(), seq, "overlapping")
    mine = np.array([cf.label_row(m, seq, table, pesm.Alphabet(), 1) for m in muts])
    assert np.abs(mine - want).max() < 2e-5
    assert Counting.windows == rb.wt_marginals_windows(n_tok, "overlapping") == {1025: 2, 1537: 2, 1538: 3, 2048: 4, 3000: 5, 3427: 6}[n_tok]


class _FakeEsmScorer:
    """Device stand-in for the runner's test seam: a deterministic score per (checkpoint, mutant)."""

    def __init__(self, location):
        self.salt = sum(map(ord, os.path.basename(location)))

    def score(self, seq, mutants, offset):
        return np.array([((self.salt * 31 + len(seq) * 7 + sum(map(ord, m))) % 1000) / 37.0 - 13.0 for m in mutants])

    def close(self):
        pass


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_reference_merge_script_consumes_the_runners_csvs(tmp_path, monkeypatch):
    """The CONSUMER side of the drop-in boundary: the reference's own proteingym/merge.py, with the registry entries of
    its own config.json (key column, score column, folder, directionality), reads the CSVs that run_benchmark (ESM-1v x 5 +
    ensemble; ESM2-3B) and run_sharded tranception write -- device calls replaced by stand-ins, everything that shapes the files is
    the product's -- and merges every model column onto the assay, one row per mutant."""
    import importlib.util
    import json
    import sys
    from proteingym_amd import run_benchmark as rb, run_sharded, synthetic, tranception as ptr
    from test_dist_cpu import _fake_tranception
    # one assay in ProteinGym's own file layout
    seq, muts, score = synthetic.random_assay(seed=3, L=50, n_single=40, n_multi=12)
    dms = tmp_path / "dms"
    dms.mkdir()
    assay = pd.DataFrame({"mutant": muts, "mutated_sequence": [ptr.get_mutated_sequence(seq, m) for m in muts], "DMS_score": score,
                          "DMS_score_bin": (score > 0).astype(int)})
    assay.to_csv(dms / "TOY_A.csv", index=False)
    pd.DataFrame([{"DMS_id": "TOY_A", "DMS_filename": "TOY_A.csv", "target_seq": seq, "DMS_total_number_mutants": len(muts),
                   "MSA_filename": "x.a2m", "MSA_start": 1, "MSA_end": len(seq), "weight_file_name": "x.npy"}]).to_csv(tmp_path / "ref.csv", index=False)
    registry = json.load(open("/root/reference/config.json"))["model_list_zero_shot_substitutions_DMS"]
    models = {k: registry[k] for k in ("ESM1v_single", "ESM1v_ensemble", "ESM2_3B", "Tranception_L_no_retrieval")}
    scores = tmp_path / "scores"
    # ESM-1v: five checkpoints named like the released files -> <scores>/ESM1v ; ESM2-3B -> <scores>/ESM2/3B
    stems = [f"esm1v_t33_650M_UR90S_{k}" for k in range(1, 6)]
    common = ["--dms_mapping", str(tmp_path / "ref.csv"), "--dms-input", str(dms)]
    rb.main(rb.create_parser().parse_args(["--model-location", *[f"/ckpt/{s}.pt" for s in stems], "--model_type", "ESM1v", *common,
                                           "--dms-output", str(scores / models["ESM1v_ensemble"]["location"])]), make_model=_FakeEsmScorer)
    rb.main(rb.create_parser().parse_args(["--model-location", "/ckpt/esm2_t36_3B_UR50D.pt", "--model_type", "ESM2", *common,
                                           "--dms-output", str(scores / models["ESM2_3B"]["location"])]), make_model=_FakeEsmScorer)
    run_sharded.main(["tranception", "--", "--checkpoint", "fake", "--DMS_reference_file_path", str(tmp_path / "ref.csv"),
                      "--DMS_data_folder", str(dms), "--output_scores_folder", str(scores / models["Tranception_L_no_retrieval"]["location"])],
                     make_model=_fake_tranception)
    cfg = {"model_list_zero_shot_substitutions_DMS": models}
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    spec = importlib.util.spec_from_file_location("pg_reference_merge", "/root/reference/proteingym/merge.py")
    merge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(merge)
    monkeypatch.setattr(sys, "argv", ["merge.py", "--DMS_assays_location", str(dms), "--model_scores_location", str(scores),
                                      "--DMS_reference_file", str(tmp_path / "ref.csv"), "--config_file", str(tmp_path / "config.json")])
    merge.main()
    merged = pd.read_csv(scores / "merged_scores" / "TOY_A.csv")
    assert len(merged) == len(assay) and list(merged["mutant"]) == muts
    ours = pd.read_csv(scores / "ESM1v" / "TOY_A.csv")
    assert list(ours.columns) == list(assay.columns) + stems + ["Ensemble_ESM1v"]
    assert np.allclose(merged["ESM1v_ensemble"], ours["Ensemble_ESM1v"]) and np.allclose(merged["ESM1v_single"], ours[stems[0]])
    assert np.allclose(merged["ESM2_3B"], pd.read_csv(scores / "ESM2" / "3B" / "TOY_A.csv")["esm2_t36_3B_UR50D"])
    tr = pd.read_csv(scores / "Tranception_no_retrieval" / "Tranception_L" / "TOY_A.csv").drop_duplicates("mutated_sequence").set_index("mutated_sequence")
    assert np.allclose(merged["Tranception_L_no_retrieval"], tr.loc[merged["mutated_sequence"], "avg_score"].to_numpy())
    assert not merged[list(models)].isna().any().any()


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_reference_merge_script_consumes_the_indel_csvs(tmp_path, monkeypatch):
    """Same for the indel benchmark (merge.py --mutation_type indels; key mutated_sequence): run_sharded tranception --indel_mode."""
    import importlib.util
    import json
    import sys
    from proteingym_amd import run_sharded, synthetic
    from test_dist_cpu import _fake_tranception
    wt, lib_ = synthetic.random_indel_library(4, 48, 30)
    dms = tmp_path / "dms"
    dms.mkdir()
    assay = pd.DataFrame({"mutant": lib_, "mutated_sequence": lib_, "DMS_score": np.linspace(-1, 1, len(lib_)), "DMS_score_bin": [0, 1] * (len(lib_) // 2)})
    assay.to_csv(dms / "TOY_I.csv", index=False)
    pd.DataFrame([{"DMS_id": "TOY_I", "DMS_filename": "TOY_I.csv", "target_seq": wt, "DMS_total_number_mutants": len(lib_)}]).to_csv(tmp_path / "ref.csv", index=False)
    registry = json.load(open("/root/reference/config.json"))["model_list_zero_shot_indels_DMS"]
    models = {"Tranception_L_no_retrieval": registry["Tranception_L_no_retrieval"]}
    scores = tmp_path / "scores"
    run_sharded.main(["tranception", "--", "--checkpoint", "fake", "--DMS_reference_file_path", str(tmp_path / "ref.csv"), "--DMS_data_folder", str(dms),
                      "--output_scores_folder", str(scores / models["Tranception_L_no_retrieval"]["location"]), "--indel_mode"], make_model=_fake_tranception)
    json.dump({"model_list_zero_shot_indels_DMS": models}, open(tmp_path / "config.json", "w"))
    spec = importlib.util.spec_from_file_location("pg_reference_merge_indels", "/root/reference/proteingym/merge.py")
    merge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(merge)
    monkeypatch.setattr(sys, "argv", ["merge.py", "--DMS_assays_location", str(dms), "--model_scores_location", str(scores), "--mutation_type", "indels",
                                      "--DMS_reference_file", str(tmp_path / "ref.csv"), "--config_file", str(tmp_path / "config.json")])
    merge.main()
    merged = pd.read_csv(scores / "merged_scores" / "TOY_I.csv")
    tr = pd.read_csv(scores / "Tranception_no_retrieval" / "Tranception_L" / "TOY_I.csv").drop_duplicates("mutated_sequence").set_index("mutated_sequence")
    assert len(merged) == len(assay) and not merged["Tranception_L_no_retrieval"].isna().any()
    assert np.allclose(merged["Tranception_L_no_retrieval"], tr.loc[merged["mutated_sequence"], "avg_score"].to_numpy())


class _GoldenCliScorer:
    """Stands in for the device with the scores the UNMODIFIED reference CLI wrote for the toy checkpoints (tests/golden/golden_esm.npz,
    keys cli/<stem> and cli_long/<stem>): the CSVs run_benchmark writes then carry the reference's own numbers in the product's files."""
    golden = None

    def __init__(self, location):
        self.stem = os.path.splitext(os.path.basename(location))[0]

    def score(self, seq, mutants, offset):
        key = ("cli_long/" if len(seq) > 1000 else "cli/") + self.stem
        v = np.asarray(self.golden[key], dtype=np.float64)
        assert len(v) == len(mutants)
        return v

    def close(self):
        pass


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_reference_performance_script_consumes_the_merged_scores(tmp_path, monkeypatch, golden_dir):
    """The LAST consumer of the boundary (north star: "drops in under performance_DMS_benchmarks.py"): run_benchmark CSVs (ESM-1v
    single + ensemble column, ESM2) of two toy assays -> the reference's merge.py -> the reference's performance_DMS_benchmarks.py
    (performance_DMS_benchmarks.py:181-195 reads merged_scores/<DMS_id>.csv and the registry's score columns).  The per-assay Spearman
    table it writes must equal scipy's spearmanr of the reference CLI's golden scores against DMS_score, at the 3 decimals the script
    keeps (and the unrounded values agree to 1e-12 with what the product's CSVs hold).  tests/test_gpu_esm.py shows the device scores
    give the same Spearman to 4 dp as these golden CLI scores.  The script's bootstrap (10 000 resamples) is cut to 20: only its
    standard-error column depends on it."""
    import importlib.util
    import json
    import shutil
    import sys
    from scipy.stats import spearmanr
    from proteingym_amd import run_benchmark as rb
    g = np.load(os.path.join(golden_dir, "golden_esm.npz"), allow_pickle=True)
    _GoldenCliScorer.golden = g
    dms = tmp_path / "dms"
    dms.mkdir()
    # four assays (the script's summary wants every taxon and every alignment-depth class present): the short and the long toy
    # assay under two names each
    layout = [("TOY_A", "TOY_DMS.csv", "seq", "Activity", "medium", "Human"), ("TOY_B", "TOY_LONG_DMS.csv", "seq_long", "Stability", "low", "Virus"),
              ("TOY_C", "TOY_DMS.csv", "seq", "Binding", "high", "Eukaryote"), ("TOY_D", "TOY_LONG_DMS.csv", "seq_long", "Expression", "medium", "Prokaryote")]
    ref_rows = []
    for dms_id, src, seq_key, sel, neff, taxon in layout:
        shutil.copy(os.path.join(golden_dir, src), dms / f"{dms_id}.csv")
        ref_rows.append(dict(DMS_id=dms_id, DMS_filename=f"{dms_id}.csv", target_seq=str(g[seq_key]), DMS_total_number_mutants=len(pd.read_csv(dms / f"{dms_id}.csv")),
                             UniProt_ID=dms_id + "_X", coarse_selection_type=sel, MSA_Neff_L_category=neff, taxon=taxon, MSA_start=1, MSA_end=len(str(g[seq_key]))))
    pd.DataFrame(ref_rows).to_csv(tmp_path / "ref.csv", index=False)
    registry = json.load(open("/root/reference/config.json"))["model_list_zero_shot_substitutions_DMS"]
    models = {k: dict(registry[k]) for k in ("ESM1v_single", "ESM1v_ensemble", "ESM2_3B")}
    models["ESM1v_single"]["input_score_name"] = "esm1v_toy_1"          # the toy checkpoints' file stems
    models["ESM2_3B"]["input_score_name"] = "esm2_toy"
    scores = tmp_path / "scores"
    common = ["--dms_mapping", str(tmp_path / "ref.csv"), "--dms-input", str(dms)]
    rb.main(rb.create_parser().parse_args(["--model-location", "/ckpt/esm1v_toy_1.pt", "--model_type", "ESM1v", *common,
                                           "--dms-output", str(scores / models["ESM1v_ensemble"]["location"])]), make_model=_GoldenCliScorer)
    rb.main(rb.create_parser().parse_args(["--model-location", "/ckpt/esm2_toy.pt", "--model_type", "ESM2", *common,
                                           "--dms-output", str(scores / models["ESM2_3B"]["location"])]), make_model=_GoldenCliScorer)
    json.dump({"model_list_zero_shot_substitutions_DMS": models}, open(tmp_path / "config.json", "w"))

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    merge = load("pg_reference_merge_perf", "/root/reference/proteingym/merge.py")
    monkeypatch.setattr(sys, "argv", ["merge.py", "--DMS_assays_location", str(dms), "--model_scores_location", str(scores),
                                      "--DMS_reference_file", str(tmp_path / "ref.csv"), "--config_file", str(tmp_path / "config.json")])
    merge.main()
    perf = load("pg_reference_performance", "/root/reference/proteingym/performance_DMS_benchmarks.py")
    slow = perf.compute_bootstrap_standard_error_functional_categories
    monkeypatch.setattr(perf, "compute_bootstrap_standard_error_functional_categories", lambda df, number_assay_reshuffle=20: slow(df, 20))
    outdir = tmp_path / "performance"
    monkeypatch.setattr(sys, "argv", ["performance_DMS_benchmarks.py", "--input_scoring_files_folder", str(scores / "merged_scores"),
                                      "--output_performance_file_folder", str(outdir), "--DMS_reference_file_path", str(tmp_path / "ref.csv"),
                                      "--DMS_data_folder", str(dms), "--config_file", str(tmp_path / "config.json")])
    perf.main()
    table = pd.read_csv(outdir / "Spearman" / "DMS_substitutions_Spearman_DMS_level.csv", index_col="DMS ID")
    clean = json.load(open("/root/reference/proteingym/constants.json"))["clean_names"]
    short = {"ESM1v_single": g["cli/esm1v_toy_1"], "ESM1v_ensemble": g["cli/esm1v_toy_1"], "ESM2_3B": g["cli/esm2_toy"]}
    long_ = {"ESM1v_single": g["cli_long/esm1v_toy_1"], "ESM1v_ensemble": g["cli_long/esm1v_toy_1"], "ESM2_3B": g["cli_long/esm2_toy"]}
    want = {"TOY_A": short, "TOY_B": long_, "TOY_C": short, "TOY_D": long_}
    for dms_id, by_model in want.items():
        dms_score = pd.read_csv(dms / f"{dms_id}.csv")["DMS_score"]
        merged = pd.read_csv(scores / "merged_scores" / f"{dms_id}.csv", float_precision="round_trip")
        assert int(table.loc[dms_id, clean.get("number_mutants", "number_mutants")]) == len(dms_score)
        for model, cli_scores in by_model.items():
            rho = spearmanr(dms_score, np.asarray(cli_scores, dtype=np.float64))[0]
            assert abs(spearmanr(merged["DMS_score"], merged[model])[0] - rho) < 1e-12          # what the product's files hold
            assert float(table.loc[dms_id, clean.get(model, model)]) == round(rho, 3)            # what the reference's script reports
    for metric in ("AUC", "MCC", "NDCG", "Top_recall"):
        assert os.path.exists(outdir / metric / f"DMS_substitutions_{metric}_DMS_level.csv")
    summary = [f for f in os.listdir(outdir / "Spearman") if f.startswith("Summary")]
    assert summary, os.listdir(outdir / "Spearman")


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_reference_performance_script_consumes_the_indel_csvs(tmp_path, monkeypatch):
    """The same for the indel benchmark (`performance_DMS_benchmarks.py --indel_mode`, key mutated_sequence): four toy indel libraries
    scored by run_sharded tranception --indel_mode (device stand-in), merged by the reference's merge.py --mutation_type indels, then
    the reference's performance script: its per-assay Spearman equals scipy's on the runner's own scores."""
    import importlib.util
    import json
    import sys
    from scipy.stats import spearmanr
    from proteingym_amd import run_sharded, synthetic
    from test_dist_cpu import _fake_tranception
    dms = tmp_path / "dms"
    dms.mkdir()
    ref_rows = []
    layout = [("IND_A", 48, 40, "Activity", "medium", "Human"), ("IND_B", 60, 36, "Stability", "low", "Virus"),
              ("IND_C", 52, 30, "Binding", "high", "Eukaryote"), ("IND_D", 45, 44, "Expression", "medium", "Prokaryote")]
    rng = np.random.default_rng(12)
    for k, (dms_id, L, n, sel, neff, taxon) in enumerate(layout):
        wt, lib_ = synthetic.random_indel_library(20 + k, L, n)
        score = rng.standard_normal(len(lib_))
        pd.DataFrame({"mutant": lib_, "mutated_sequence": lib_, "DMS_score": score, "DMS_score_bin": (score > 0).astype(int)}).to_csv(dms / f"{dms_id}.csv", index=False)
        ref_rows.append(dict(DMS_id=dms_id, DMS_filename=f"{dms_id}.csv", target_seq=wt, DMS_total_number_mutants=len(lib_), UniProt_ID=dms_id + "_X",
                             coarse_selection_type=sel, MSA_Neff_L_category=neff, taxon=taxon))
    pd.DataFrame(ref_rows).to_csv(tmp_path / "ref.csv", index=False)
    registry = json.load(open("/root/reference/config.json"))["model_list_zero_shot_indels_DMS"]
    models = {"Tranception_L_no_retrieval": registry["Tranception_L_no_retrieval"]}
    scores = tmp_path / "scores"
    run_sharded.main(["tranception", "--", "--checkpoint", "fake", "--DMS_reference_file_path", str(tmp_path / "ref.csv"), "--DMS_data_folder", str(dms),
                      "--output_scores_folder", str(scores / models["Tranception_L_no_retrieval"]["location"]), "--indel_mode"], make_model=_fake_tranception)
    json.dump({"model_list_zero_shot_indels_DMS": models}, open(tmp_path / "config.json", "w"))

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    merge = load("pg_reference_merge_perf_indels", "/root/reference/proteingym/merge.py")
    monkeypatch.setattr(sys, "argv", ["merge.py", "--DMS_assays_location", str(dms), "--model_scores_location", str(scores), "--mutation_type", "indels",
                                      "--DMS_reference_file", str(tmp_path / "ref.csv"), "--config_file", str(tmp_path / "config.json")])
    merge.main()
    perf = load("pg_reference_performance_indels", "/root/reference/proteingym/performance_DMS_benchmarks.py")
    slow = perf.compute_bootstrap_standard_error_functional_categories
    monkeypatch.setattr(perf, "compute_bootstrap_standard_error_functional_categories", lambda df, number_assay_reshuffle=20: slow(df, 20))
    outdir = tmp_path / "performance"
    monkeypatch.setattr(sys, "argv", ["performance_DMS_benchmarks.py", "--input_scoring_files_folder", str(scores / "merged_scores"),
                                      "--output_performance_file_folder", str(outdir), "--DMS_reference_file_path", str(tmp_path / "ref.csv"),
                                      "--DMS_data_folder", str(dms), "--config_file", str(tmp_path / "config.json"), "--indel_mode"])
    perf.main()
    table = pd.read_csv(outdir / "Spearman" / "DMS_indels_Spearman_DMS_level.csv", index_col="DMS ID")
    clean = json.load(open("/root/reference/proteingym/constants.json"))["clean_names"]
    col = clean.get("Tranception_L_no_retrieval", "Tranception_L_no_retrieval")
    for dms_id, *_ in layout:
        ours = pd.read_csv(scores / "Tranception_no_retrieval" / "Tranception_L" / f"{dms_id}.csv").drop_duplicates("mutated_sequence").set_index("mutated_sequence")
        assay = pd.read_csv(dms / f"{dms_id}.csv")
        rho = spearmanr(assay["DMS_score"], ours.loc[assay["mutated_sequence"], "avg_score"].to_numpy())[0]
        assert float(table.loc[dms_id, col]) == round(rho, 3), (dms_id, table.loc[dms_id, col], rho)


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("kwargs", [
    {},
    {"threshold_sequence_frac_gaps": 0.2, "threshold_focus_cols_frac_gaps": 0.3},
    {"preprocess_MSA": False},
    {"remove_sequences_with_indeterminate_AA_in_focus_cols": False},
    {"preprocess_MSA": False, "remove_sequences_with_indeterminate_AA_in_focus_cols": False},
])
def test_alignment_preprocessing_vs_live_reference(tmp_path, kwargs):
    """alignment.FocusAlignment (byte-matrix numpy) against the reference's own ``MSA_processing`` (string / pandas code,
    tranception/utils/msa_utils.py:230-340) on an alignment with insert columns, '.', gappy rows and columns, X / B / Z residues and
    a sequence spread over several lines -- every attribute a caller can read, for the reference's switches on and off."""
    from proteingym_amd import tranception as ptr, msa_transformer as pmsa
    rh.load_reference_tranception()
    from tranception.utils import msa_utils
    rng = np.random.default_rng(11)
    aa = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    width, focus = 60, None
    lines = []
    col_kind = rng.choice(3, size=width, p=[0.75, 0.15, 0.10])               # 0 match column, 1 insert column (focus '.' / lower), 2 gappy
    for i in range(40):
        row = aa[rng.integers(0, 20, width)].copy()
        if i == 0:
            row[col_kind == 1] = "-" if kwargs.get("preprocess_MSA", True) else "."
            if not kwargs.get("preprocess_MSA", True):
                low = rng.random(width) < 0.1
                row = np.where(low & (col_kind == 0), np.char.lower(row), row)
        else:
            row[rng.random(width) < (0.9 if i % 7 == 3 else 0.12)] = "-"
            row[(col_kind == 2) & (rng.random(width) < 0.8)] = "-"
            ins = col_kind == 1
            row[ins] = np.where(rng.random(ins.sum()) < 0.5, ".", np.char.lower(row[ins]))
            if i % 9 == 4:
                row[rng.integers(0, width)] = "X"
            if i % 11 == 5:
                row[rng.integers(0, width)] = "b"
        s = "".join(row)
        lines.append(f">seq{i}/10-{9 + width}" if i else ">FOCUS/10-69")
        lines += [s[:25], s[25:]] if i % 3 == 1 else [s]
    a2m = tmp_path / "t.a2m"
    a2m.write_text("\n".join(lines) + "\n")
    ref = msa_utils.MSA_processing(MSA_location=str(a2m), use_weights=False, **kwargs)
    for mine in (ptr.MSA_processing(MSA_location=str(a2m), use_weights=False, **kwargs),
                 pmsa.MSA_processing(MSA_location=str(a2m), use_weights=False, **kwargs)):
        assert mine.focus_seq_name == ref.focus_seq_name and mine.focus_seq == ref.focus_seq
        assert list(mine.focus_cols) == list(ref.focus_cols) and mine.seq_len == ref.seq_len
        assert "".join(mine.focus_seq_trimmed) == "".join(ref.focus_seq_trimmed)
        assert dict(mine.raw_seq_name_to_sequence) == dict(ref.raw_seq_name_to_sequence)
        assert list(mine.seq_name_to_sequence) == list(ref.seq_name_to_sequence)
        for n in ref.seq_name_to_sequence:
            assert "".join(mine.seq_name_to_sequence[n]) == "".join(ref.seq_name_to_sequence[n]), n
        assert mine.num_sequences == ref.num_sequences and np.array_equal(mine.weights, ref.weights) and mine.Neff == ref.Neff
        assert list(mine.seq_name_to_weight) == list(ref.seq_name_to_weight)
    tr = ptr.MSA_processing(MSA_location=str(a2m), use_weights=False, **kwargs)
    assert (tr.focus_start_loc, tr.focus_stop_loc) == (ref.focus_start_loc, ref.focus_stop_loc)
    assert tr.uniprot_focus_col_to_wt_aa_dict == ref.uniprot_focus_col_to_wt_aa_dict
    assert tr.uniprot_focus_col_to_focus_idx == ref.uniprot_focus_col_to_focus_idx
    onehot = np.zeros((tr.num_sequences, tr.seq_len, 20))
    i, j = np.nonzero(tr.encoded >= 0)
    onehot[i, j, tr.encoded[i, j]] = 1.0
    assert np.array_equal(onehot, ref.one_hot_encoding)


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("weighted", [False, True])
def test_msa_prior_in_blocks_equals_live_reference_bitwise(tmp_path, weighted):
    """get_msa_prior reduces the alignment in blocks of sequences (bounded memory); the bits must be the reference's, which reduces
    one [sequences, width, V] array (msa_utils.py:63-138) -- 300 sequences in 24 blocks, with and without EVE weights."""
    from proteingym_amd import tranception as ptr
    rh.load_reference_tranception()
    from tranception.utils import msa_utils
    rng = np.random.default_rng(5)
    aa = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    width = 48
    focus = aa[rng.integers(0, 20, width)]
    lines = [">FOCUS/3-50", "".join(focus)]
    for i in range(1, 300):
        row = np.where(rng.random(width) < (0.95 if i % 13 == 0 else 0.4), aa[rng.integers(0, 20, width)], focus)
        row[rng.random(width) < 0.15] = "-"
        if i % 17 == 0:
            row[rng.integers(0, width)] = "X"
        lines += [f">s{i}", "".join(row)]
    a2m = tmp_path / "p.a2m"
    a2m.write_text("\n".join(lines) + "\n")
    wfile = None
    if weighted:
        wfile = str(tmp_path / "w.npy")
        msa_utils.MSA_processing(MSA_location=str(a2m), use_weights=True, weights_location=wfile)      # the reference computes and saves
    want = msa_utils.get_msa_prior(MSA_data_file=str(a2m), MSA_weight_file_name=wfile, MSA_start=2, MSA_end=50, len_target_seq=60,
                                   vocab=ptr.VOCAB, verbose=False)
    got = ptr.get_msa_prior(str(a2m), wfile, 2, 50, 60, block_bytes=13 * width * 25 * 8)
    assert np.array_equal(got, want)
    assert np.array_equal(ptr.get_msa_prior(str(a2m), wfile, 2, 50, 60), want)


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_alignment_reader_oddities_vs_live_reference(tmp_path):
    """alignment.read_records against the reference's two readers (msa_utils.py:28-43 ``process_msa_data``; the loop at the top of
    ``MSA_processing.gen_alignment``) on what real a2m files occasionally hold: a sequence over several lines, a repeated header, blank
    lines, trailing blanks, lower case and '.', a header without a sequence."""
    from proteingym_amd import tranception as ptr, alignment
    rh.load_reference_tranception()
    from tranception.utils import msa_utils
    text = (">FOCUS/1-8\nACDE\nFGHI  \n\n>two\nac.eF-HI\n>empty\n>three\nACDEFGHI\n>two\nKLMNPQRS\n>four \nA-DE\n\nFG-I\n")
    p = tmp_path / "odd.a2m"
    p.write_text(text)
    want = msa_utils.process_msa_data(str(p))
    got = ptr.process_msa_data(str(p))
    assert list(got.items()) == list(want.items())
    first, raw = alignment.read_records(str(p))
    ref = msa_utils.MSA_processing.__new__(msa_utils.MSA_processing)           # only the reading loop is wanted: drive gen_alignment's own state
    ref.MSA_location, ref.alphabet, ref.preprocess_MSA, ref.use_weights = str(p), "ACDEFGHIKLMNPQRSTVWY", False, False
    ref.remove_sequences_with_indeterminate_AA_in_focus_cols, ref.weights_location, ref.theta = False, None, 0.2
    ref.gen_alignment()                                                          # no pre-processing: the records as read
    assert first == ref.focus_seq_name
    assert list(raw.items()) == list(ref.raw_seq_name_to_sequence.items())
    with pytest.raises(ValueError, match="differ in length"):                  # '>two' was extended by its second record: ragged
        alignment.FocusAlignment(str(p), preprocess=False)


# ---- Tranception: indel scoring WITH retrieval (the aligner is the user's executable; here a deterministic stand-in) ------------------
STAND_IN_ALIGNER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stand_in_clustalo.py")


def _indel_retrieval(to, golden_dir, a2m, ms, me, L, work):
    prior = to.get_msa_prior(os.path.join(golden_dir, a2m), ms, me, L)
    with np.errstate(divide="ignore"):
        log_prior = torch.log(torch.tensor(prior).float()).numpy()
    return dict(log_prior=log_prior, MSA_start=ms, MSA_end=me, weight=0.6,
                aligner=to.clustal_aligner(os.path.join(golden_dir, a2m), STAND_IN_ALIGNER, str(work)))


def test_tranception_oracle_indels_with_retrieval_reproduces_golden(golden_dir, tmp_path):
    """tests/golden/make_golden_tranception_indel_retrieval.py ran the unmodified reference (model_pytorch.py:794-840,
    msa_utils.py:141-192) with the stand-in aligner: the oracle's restatement -- aligner file protocol, the walk that edits the prior, the
    zero-row fusion rule, both directions, wild-type delta -- gives the same three columns, gets the same aligned rows from the stand-in,
    and fails the way the reference fails on an alignment that does not cover the scored window."""
    from oracle import tranception_oracle as to
    g = np.load(os.path.join(golden_dir, "golden_tranception_indel_retrieval.npz"))
    seq = str(g["seq"])
    cfg, W = to.load_checkpoint(os.path.join(golden_dir, "Tranception_toy"))
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_INDEL_RETRIEVAL_DMS.csv"))
    retr = _indel_retrieval(to, golden_dir, "TOY_MSA_INDEL_FULL.a2m", 0, len(seq), len(seq), tmp_path / "full")
    for k, s in enumerate(df["mutated_sequence"]):
        assert list(retr["aligner"](s)) == [str(v) for v in g[f"aligned/{k}"]], k
    r = to.score_mutants(cfg, W, df[["mutant", "mutated_sequence"]], seq, retrieval=retr, indel_mode=True)
    key = r["mutated_sequence"].fillna(r["mutant"]) if "mutant" in r else r["mutated_sequence"]
    m = pd.merge(df[["mutated_sequence"]], r.assign(key=key), left_on="mutated_sequence", right_on="key", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(m[c].to_numpy(dtype=np.float64) - g[f"full/{c}"]).max() < 2e-5, c
    assert float(m["avg_score"][0]) == 0.0                                       # the wild type is in the library: its zero row
    assert str(g["partial_alignment"]).startswith("IndexError")
    part = _indel_retrieval(to, golden_dir, "TOY_MSA.a2m", 10, 60, len(seq), tmp_path / "part")
    with pytest.raises(IndexError):
        to.score_mutants(cfg, W, df[["mutant", "mutated_sequence"]].iloc[:3], seq, retrieval=part, indel_mode=True)


def test_update_prior_indel_walk():
    """The walk of msa_utils.py:174-191 on hand-made rows: a deleted residue drops its row, an inserted one gets a zero row, 'both gaps'
    columns are skipped but still advance the insertion index (the reference's own indexing), MSA_end follows the row count; a mask
    that does not fit leaves the prior as edited so far and MSA_end untouched (the reference's bare except)."""
    from oracle import tranception_oracle as to
    V = 5
    prior = torch.arange(1, 7, dtype=torch.float32).view(6, 1).repeat(1, V)            # rows 1..6
    out, ms, me = to.update_prior_indel(prior, 0, 6, "AB-DEFX", "ABCDEF-")              # C deleted (col 2), X inserted after F (col 6)
    assert (ms, me) == (0, 6) and out[:, 0].tolist() == [1, 2, 4, 5, 6, 0]
    out, ms, me = to.update_prior_indel(prior, 0, 6, "A-BXCDEF", "A-B-CDEF")            # col 1 both gaps: skipped; X at col 3 -> zero row at index 3
    assert out[:, 0].tolist() == [1, 2, 3, 0, 4, 5, 6] and me == 7
    out, ms, me = to.update_prior_indel(prior, 0, 6, "ABC", "ABC")                      # 3 mask entries for 6 rows: IndexError inside, swallowed
    assert out[:, 0].tolist() == [1, 2, 3, 4, 5, 6] and me == 6


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_update_prior_indel_vs_live_reference(tmp_path):
    """The same walk through the reference's own function (msa_utils.update_retrieved_MSA_log_prior_indel driven with a stand-in model object
    and the stand-in aligner) on random indels of a random family, incl. an alignment whose reference row has gaps."""
    from oracle import tranception_oracle as to
    rh.load_reference_tranception()
    from tranception.utils import msa_utils
    import types
    rng = np.random.default_rng(8)
    aa = list("ACDEFGHIKLMNPQRSTVWY")
    for case in range(3):
        L = 40
        ref = "".join(rng.choice(aa, size=L))
        rows = [ref if case < 2 else ref[:10] + "--" + ref[12:]]                          # case 2: gaps in the reference row itself
        for _ in range(6):
            s = list(ref)
            for p_ in rng.choice(L, size=6, replace=False):
                s[p_] = rng.choice(aa + ["-"])
            rows.append("".join(s))
        folder = tmp_path / f"fam{case}"
        folder.mkdir()
        a2m = folder / "fam.a2m"
        a2m.write_text("".join(f">s{i}/1-{L}\n{r}\n" for i, r in enumerate(rows)))
        model = types.SimpleNamespace(MSA_folder=str(folder), MSA_filename=str(a2m),
                                      config=types.SimpleNamespace(clustal_omega_location=STAND_IN_ALIGNER))
        prior = torch.log(torch.rand(L, 25, generator=torch.Generator().manual_seed(case)))
        aligner = to.clustal_aligner(str(a2m), STAND_IN_ALIGNER, str(tmp_path / f"work{case}"))
        for trial in range(6):
            s = list(ref.replace("-", ""))
            for _ in range(int(rng.integers(1, 4))):
                p_ = int(rng.integers(1, len(s) - 1))
                if rng.random() < 0.5:
                    del s[p_]
                else:
                    s.insert(p_, rng.choice(aa))
            s = "".join(s)
            want, ws, we = msa_utils.update_retrieved_MSA_log_prior_indel(model, prior.clone(), 0, L, s, "hash")
            got, gs, ge = to.update_prior_indel(prior.clone(), 0, L, *aligner(s))
            assert (gs, ge) == (ws, we) and got.shape == want.shape and torch.equal(got, want), (case, trial)


class _OracleBackedDevice:
    """``pgmi_tr_sequence_loglik`` with the ABI's semantics (include/pgmi.h: per sequence ONE contiguous run of logit rows
    [a0, a0 + n) fused with log_prior rows row0 + i, or row0 + n - 1 - i when flipped) computed by the oracle's forward -- lets
    the product's indel-with-retrieval host logic (re-indexed prior, runs of positions, sum over copies) execute on CPU."""

    def __init__(self, cfg, W):
        self.cfg, self.W = cfg, W
        self.calls = []

    def pgmi_tr_sequence_loglik(self, handle, tokens, lens, B, T, log_prior, P, a0, row0, count, flip, alpha, out):
        from oracle import tranception_oracle as to
        ids = np.ctypeslib.as_array(tokens, shape=(B, T)).astype(np.int64)
        ln = np.ctypeslib.as_array(lens, shape=(B,))
        res = np.ctypeslib.as_array(out, shape=(B,))
        mask = (np.arange(T)[None, :] < ln[:, None]).astype(np.int64)
        with torch.no_grad():
            lp = torch.log_softmax(to.forward_logits(self.cfg, self.W, ids, mask)[:, :-1, :], dim=-1)
        if log_prior:
            prior = torch.as_tensor(np.ctypeslib.as_array(log_prior, shape=(P, lp.shape[-1])).copy())
            A0, R0, N, F = (np.ctypeslib.as_array(p_, shape=(B,)) for p_ in (a0, row0, count, flip))
            self.calls.append((B, N.tolist()))
            for b in range(B):
                if N[b] > 0:
                    rows = prior[R0[b]:R0[b] + N[b]]
                    rows = torch.flip(rows, dims=(0,)) if F[b] else rows
                    lp[b, A0[b]:A0[b] + N[b]] = (1 - alpha) * lp[b, A0[b]:A0[b] + N[b]] + alpha * rows
        ll = torch.gather(lp, 2, torch.as_tensor(ids[:, 1:]).unsqueeze(-1)).squeeze(-1)
        res[:] = (ll * torch.as_tensor(mask[:, 1:]).to(ll.dtype)).sum(1).numpy()
        return 0

    def pgmi_tr_sequence_loglik_shared(self, handle, tokens, ref, B, T, log_prior, P, a0, row0, count, flip, alpha, out, token_lp, rows):
        """The shared-prefix entry's CONTRACT (include/pgmi.h): every sequence has T tokens, ref[b] is a root of this call whose
        prefix b shares -- checked here -- and the values are those of the unshared call (the oracle forwards in full)."""
        ids = np.ctypeslib.as_array(tokens, shape=(B, T))
        r = np.ctypeslib.as_array(ref, shape=(B,))
        assert ((r >= 0) & (r < B)).all() and (r[r] == r).all()
        assert (ids[np.arange(B), 0] == ids[r, 0]).all() and (ids != 3).all()            # the [CLS] is shared; no [PAD]
        first_diff = np.array([T if (ids[b] == ids[r[b]]).all() else int(np.argmax(ids[b] != ids[r[b]])) for b in range(B)])
        self.shared.append((B, int((r != np.arange(B)).sum()), first_diff[r != np.arange(B)].tolist()))
        lens = (C.c_int32 * B)(*([T] * B))
        rc = self.pgmi_tr_sequence_loglik(handle, tokens, lens, B, T, log_prior, P, a0, row0, count, flip, alpha, out)
        if rows:
            rows[0] = int(sum(T - min(p, T - 1) for p in first_diff[r != np.arange(B)])) + T * int((r == np.arange(B)).sum())
        return rc

    shared = []


def test_tranception_product_indels_with_retrieval_host_logic_on_cpu(golden_dir, tmp_path, monkeypatch):
    """The product's side of indel scoring with retrieval (tranception.SequenceAligner, realigned_prior_rows,
    TranceptionModel._realigned_loglik, the CLI's retrieval arguments) against the unmodified reference's goldens, with the device
    reduction served by the oracle through the ABI's own semantics.  An inserted residue splits the fused positions into runs: the
    sequence then goes through the reduction once per run plus once without a prior."""
    import shutil
    from oracle import tranception_oracle as to
    from proteingym_amd import tranception as ptr, _lib, score_tranception_proteingym as cli
    g = np.load(os.path.join(golden_dir, "golden_tranception_indel_retrieval.npz"))
    seq = str(g["seq"])
    cfg, W = to.load_checkpoint(os.path.join(golden_dir, "Tranception_toy"))
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_INDEL_RETRIEVAL_DMS.csv"))
    a2m = shutil.copy(os.path.join(golden_dir, "TOY_MSA_INDEL_FULL.a2m"), tmp_path / "TOY_MSA_INDEL_FULL.a2m")   # the aligner writes next to it
    args = cli.create_parser().parse_args(["--checkpoint", "x", "--DMS_data_folder", "x", "--indel_mode", "--inference_time_retrieval",
                                           "--clustal_omega_location", STAND_IN_ALIGNER])
    retrieval = ptr.build_retrieval(cli.retrieval_arguments(args, seq, (str(a2m), None, 0, len(seq))))
    assert isinstance(retrieval["aligner"], ptr.SequenceAligner)
    for k, s in enumerate(df["mutated_sequence"]):
        assert list(retrieval["aligner"](s)) == [str(v) for v in g[f"aligned/{k}"]], k
    device = _OracleBackedDevice(cfg, W)
    monkeypatch.setattr(_lib, "load", lambda: device)
    model = object.__new__(ptr.TranceptionModel)
    model._h, model.n_ctx, model.scoring_window, model.retrieval, model.cfg = None, cfg["n_ctx"], "optimal", retrieval, cfg
    r = model.score_mutants(DMS_data=df, target_seq=seq, scoring_mirror=True, indel_mode=True)
    key = r["mutated_sequence"].fillna(r["mutant"]) if "mutant" in r else r["mutated_sequence"]
    m = pd.merge(df[["mutated_sequence"]], r.assign(key=key), left_on="mutated_sequence", right_on="key", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(m[c].to_numpy(dtype=np.float64) - g[f"full/{c}"]).max() < 3e-5, c
    assert max(b for b, _ in device.calls) >= 3 and min(b for b, _ in device.calls) == 2      # insertions inside: 2+ runs; none: 1 run + the plain copy
    with pytest.raises(ValueError, match="clustal_omega_location"):
        cli.retrieval_arguments(cli.create_parser().parse_args(["--checkpoint", "x", "--DMS_data_folder", "x", "--indel_mode",
                                                                "--inference_time_retrieval"]), seq, (str(a2m), None, 0, len(seq)))
    part = shutil.copy(os.path.join(golden_dir, "TOY_MSA.a2m"), tmp_path / "TOY_MSA.a2m")
    model.retrieval = ptr.build_retrieval(dict(MSA_filename=str(part), MSA_weight_file_name=None, MSA_start=10, MSA_end=60,
                                               full_protein_length=len(seq), retrieval_aggregation_mode="aggregate_indel",
                                               clustal_omega_location=STAND_IN_ALIGNER))
    with pytest.raises(IndexError):                                            # the reference raises IndexError on this input too (golden)
        model.score_mutants(DMS_data=df.iloc[:3], target_seq=seq, scoring_mirror=True, indel_mode=True)


def test_run_sharded_tranception_indels_with_retrieval_in_chunks_on_cpu(golden_dir, tmp_path, monkeypatch):
    """``run_sharded tranception`` (mutant chunks, resident model, per-assay retrieval swapped in) with --indel_mode
    --inference_time_retrieval --clustal_omega_location: the assay cut into three chunks gives the unmodified reference's columns (the
    device call served by the oracle through the ABI's semantics)."""
    import shutil
    from oracle import tranception_oracle as to
    from proteingym_amd import tranception as ptr, _lib, run_sharded
    g = np.load(os.path.join(golden_dir, "golden_tranception_indel_retrieval.npz"))
    seq = str(g["seq"])
    cfg, W = to.load_checkpoint(os.path.join(golden_dir, "Tranception_toy"))
    device = _OracleBackedDevice(cfg, W)
    monkeypatch.setattr(_lib, "load", lambda: device)

    def make_model(checkpoint, dev, scoring_window):
        m = object.__new__(ptr.TranceptionModel)
        m._h, m.n_ctx, m.scoring_window, m.retrieval, m.cfg = None, cfg["n_ctx"], scoring_window, None, cfg
        m.close = lambda: None
        return m
    (tmp_path / "msa").mkdir()
    (tmp_path / "data").mkdir()
    shutil.copy(os.path.join(golden_dir, "TOY_MSA_INDEL_FULL.a2m"), tmp_path / "msa" / "TOY_MSA_INDEL_FULL.a2m")
    shutil.copy(os.path.join(golden_dir, "TOY_TRANCEPTION_INDEL_RETRIEVAL_DMS.csv"), tmp_path / "data" / "A.csv")
    pd.DataFrame({"DMS_id": ["A"], "DMS_filename": ["A.csv"], "target_seq": [seq], "MSA_filename": ["TOY_MSA_INDEL_FULL.a2m"], "MSA_start": [1],
                  "MSA_end": [len(seq)], "weight_file_name": ["none.npy"]}).to_csv(tmp_path / "ref.csv", index=False)
    run_sharded.main(["tranception", "--max-chunk-rows", "6", "--", "--checkpoint", "x", "--DMS_reference_file_path", str(tmp_path / "ref.csv"),
                      "--DMS_data_folder", str(tmp_path / "data"), "--output_scores_folder", str(tmp_path / "out"), "--indel_mode",
                      "--inference_time_retrieval", "--MSA_folder", str(tmp_path / "msa"), "--clustal_omega_location", STAND_IN_ALIGNER],
                     make_model=make_model)
    r = pd.read_csv(tmp_path / "out" / "A.csv", float_precision="round_trip")
    df = pd.read_csv(tmp_path / "data" / "A.csv")
    key = r["mutated_sequence"].fillna(r["mutant"]) if "mutant" in r else r["mutated_sequence"]
    m = pd.merge(df[["mutated_sequence"]], r.assign(key=key), left_on="mutated_sequence", right_on="key", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(m[c].to_numpy(dtype=np.float64) - g[f"full/{c}"]).max() < 3e-5, c


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_filter_msa_vs_live_reference(tmp_path, monkeypatch):
    """--filter-msa (compute_fitness.py:76-97): the reference rewrites the alignment ('.' -> '-', upper case, headers included), shells out
    to <path>/bin/hhfilter and pre-processes what comes back.  The product does the same through subprocess; with the stand-in filter
    (tests/golden/stand_in_hhfilter.py) in place of hh-suite both hand the SAME filtered file (same relative path, same bytes) to
    MSA_processing -- whose own parity is test_alignment_preprocessing_vs_live_reference."""
    import stat
    from proteingym_amd import msa_transformer as pmsa
    ref_cf = rh.load_reference()
    rng = np.random.default_rng(21)
    aa = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    width = 50
    focus = aa[rng.integers(0, 20, width)]
    lines = [">Target.prot/1-50", "".join(focus)]
    for i in range(1, 60):
        row = np.where(rng.random(width) < (0.97 if i % 5 == 0 else 0.5), focus, aa[rng.integers(0, 20, width)])
        row[rng.random(width) < (0.6 if i % 7 == 0 else 0.1)] = "-"
        lines += [f">UniRef100_x{i}.v/1-50", "".join(row)]
    root = tmp_path / "hhsuite"
    (root / "bin").mkdir(parents=True)
    exe = root / "bin" / "hhfilter"
    exe.write_text('#!/bin/sh\nexec python3 "%s" "$@"\n' % os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stand_in_hhfilter.py"))
    exe.chmod(exe.stat().st_mode | stat.S_IXUSR)
    seen = {}
    for who, fn, mod in (("ref", ref_cf.process_msa, ref_cf), ("mine", pmsa.process_msa, pmsa)):
        folder = tmp_path / who
        folder.mkdir()
        (folder / "fam.a2m").write_text("\n".join(lines) + "\n")
        import types
        monkeypatch.setattr(mod, "MSA_processing",                                                 # record what the pre-processing is given
                            lambda MSA_location, **kw: types.SimpleNamespace(focus_seq_name="recorded", MSA_location=MSA_location))
        out = fn(filename=str(folder / "fam.a2m"), weight_filename=None, filter_msa=True, path_to_hhfilter=str(root), hhfilter_min_cov=75,
                 hhfilter_max_seq_id=90, hhfilter_min_seq_id=0).MSA_location
        seen[who] = (os.path.relpath(out, folder), open(out).read())
    assert seen["mine"] == seen["ref"]
    kept = seen["ref"][1].count(">")
    assert 1 < kept < 60 and ">TARGET-PROT/1-50" in seen["ref"][1]                # filtered; headers went through tr and ucase too
    with pytest.raises(RuntimeError, match="hhfilter"):
        pmsa.process_msa(filename=str(tmp_path / "mine" / "fam.a2m"), weight_filename=None, filter_msa=True, path_to_hhfilter=str(tmp_path / "nowhere"))


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_tranception_host_helpers_vs_live_reference():
    """The column-wise helpers of tranception.py against the reference's per-string functions (scoring_utils.py:16-31,47-69): mutated
    sequences incl. repeated positions and negative indices, the two assertion texts in the reference's order of checks, windows at
    every position of proteins around the context length, and the random replacement consuming numpy's stream like the reference."""
    from proteingym_amd import synthetic, tranception as ptr
    rh.load_reference_tranception()
    from tranception.utils import scoring_utils as ref
    seq, muts, _ = synthetic.random_assay(seed=3, L=90, n_single=60, n_multi=140)
    w0 = seq[4]
    muts += [f"{w0}5A:{w0}5C", f"{seq[-1]}0G", f"{seq[9]}10W:{seq[0]}1Y:{seq[9]}10H"]        # same position twice; position 0 = index -1
    assert ptr.mutated_sequences(seq, muts) == [ref.get_mutated_sequence(seq, m) for m in muts]
    assert ptr.get_mutated_sequence(seq, muts[7], start_idx=1) == ref.get_mutated_sequence(seq, muts[7], start_idx=1)
    other = "A" if seq[2] != "A" else "C"
    for bad in ([muts[0], f"{other}3G"], [f"{seq[2]}3B", muts[1]], [f"{seq[2]}3G:{other}3G:{seq[4]}5B"], [f"{seq[2]}3B:{other}3G"]):
        with pytest.raises(AssertionError) as want:
            [ref.get_mutated_sequence(seq, m) for m in bad]
        with pytest.raises(AssertionError) as got:
            ptr.mutated_sequences(seq, bad)
        assert str(got.value) == str(want.value)
    with pytest.raises(IndexError):
        ptr.mutated_sequences(seq, [f"{seq[0]}500A"])
    for n, w in ((50, 1022), (1022, 1022), (1023, 1022), (1500, 1022), (77, 20), (78, 21)):
        pos = np.arange(n)
        want = np.array([ref.get_optimal_window(int(p), n, w) for p in pos])
        assert np.array_equal(ptr.optimal_windows(pos, n, w), want)
        assert all(ptr.get_optimal_window(int(p), n, w) == list(want[p]) for p in (0, n // 2, n - 1))
    text = "MXKBXZJAXXB"
    for ch, choices in (("X", "ACDEFGHIKLMNPQRSTVWY"), ("B", "DN"), ("J", "IL"), ("Q", "EQ")):
        np.random.seed(11)
        want = ref.sequence_replace_single(text, ch, choices)
        want_next = np.random.random()
        np.random.seed(11)
        assert ptr.sequence_replace_single(text, ch, choices) == want
        assert np.random.random() == want_next or ch == "Q"           # no occurrence: the reference still calls choice(size=0)


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_msa_prior_ragged_alignment_like_the_live_reference(tmp_path):
    """Rows shorter than the query: the reference tolerates them without the similarity filter (their tail holds no counts) and fails
    with ValueError in the filter's np.dot; rows longer than MSA_end - MSA_start with residues there: IndexError in both."""
    from proteingym_amd import tranception as ptr
    rh.load_reference_tranception()
    from tranception.utils import msa_utils
    a2m = tmp_path / "r.a2m"
    a2m.write_text(">Q/1-12\nACDEFGHIKLMN\n>a\nACDEFGHI\n>b\nAC-EFGHIKLMN\n>c\nACD\n")
    kw = dict(MSA_data_file=str(a2m), MSA_weight_file_name=None, MSA_start=0, MSA_end=12, len_target_seq=14, vocab=ptr.VOCAB)
    want = msa_utils.get_msa_prior(filter_MSA=False, verbose=False, **kw)
    assert np.array_equal(ptr.get_msa_prior(str(a2m), None, 0, 12, 14, filter_MSA=False), want)
    with pytest.raises(ValueError):
        msa_utils.get_msa_prior(filter_MSA=True, verbose=False, **kw)
    with pytest.raises(ValueError):
        ptr.get_msa_prior(str(a2m), None, 0, 12, 14, filter_MSA=True)
    with pytest.raises(IndexError):
        msa_utils.get_msa_prior(MSA_data_file=str(a2m), MSA_weight_file_name=None, MSA_start=0, MSA_end=8, len_target_seq=14, vocab=ptr.VOCAB,
                                filter_MSA=False, verbose=False)
    with pytest.raises(IndexError):
        ptr.get_msa_prior(str(a2m), None, 0, 8, 14, filter_MSA=False)


def test_tranception_product_prefix_sharing_host_logic_on_cpu(golden_dir, monkeypatch):
    """The host side of prefix-shared scoring (tranception.wild_type_rows, TranceptionModel.sequence_loglik / _local_references) through
    the product's own score_mutants with the device served by the oracle: every mutated slice names the wild type cut to ITS window as
    its root (short protein: one window; 1 100 residues: a window per mutated position; 'sliding': per chunk), roots stand for
    themselves, and the reference's goldens come out.  With share_prefix off the plain entry is used."""
    from oracle import tranception_oracle as to
    from proteingym_amd import tranception as ptr, _lib
    g = np.load(os.path.join(golden_dir, "golden_tranception.npz"))
    gm = np.load(os.path.join(golden_dir, "golden_tranception_modes.npz"))
    seq, seql = str(g["seq"]), str(g["seq_long"])
    cfg, W = to.load_checkpoint(os.path.join(golden_dir, "Tranception_toy"))
    device = _OracleBackedDevice(cfg, W)
    device.shared = []
    monkeypatch.setattr(_lib, "load", lambda: device)

    def model(window="optimal", share=True):
        m = object.__new__(ptr.TranceptionModel)
        m._h, m.n_ctx, m.scoring_window, m.retrieval, m.cfg, m.share_prefix = None, cfg["n_ctx"], window, None, cfg, share
        return m

    def check(m, df, target, gold, prefix):
        r = m.score_mutants(DMS_data=df, target_seq=target, scoring_mirror=True)
        r = pd.merge(pd.DataFrame({"mutated_sequence": [to.get_mutated_sequence(target, x) for x in df["mutant"]]}), r, on="mutated_sequence", how="left")
        for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
            assert np.abs(r[c].to_numpy() - gold[f"{prefix}/{c}"]).max() < 2e-5, (prefix, c)
    dms = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_DMS.csv"))
    dml = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_LONG_DMS.csv"))
    m = model()
    check(m, dms, seq, g, "scores")
    assert len(device.shared) == 2                                         # one call per reading direction: one length, one window
    for B, n_shared, first in device.shared:
        assert n_shared == B - 1 and min(first) >= 1                       # everything but the wild type shares; [CLS] is never the difference
    assert 0 < m.rows_forwarded < m.rows_full
    device.shared = []
    check(model(), dml, seql, g, "scores_long")                            # longer than the context: the root of a slice is the wild type of ITS window
    assert sum(n for _, n, _ in device.shared) > 0
    device.shared = []
    check(model("sliding"), dml, seql, gm, "sliding")
    assert sum(n for _, n, _ in device.shared) > 0
    device.shared = []
    m = model(share=False)
    check(m, dms, seq, g, "scores")
    assert device.shared == [] and m.rows_forwarded == m.rows_full > 0
    # no reference sequence: raw log-likelihoods, nothing to share
    device.shared = []
    r = model().score_mutants(DMS_data=pd.DataFrame({"mutated_sequence": [seq, to.get_mutated_sequence(seq, dms["mutant"][0])]}), target_seq=None)
    assert device.shared == [] and len(r) == 2


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_accept_real_weights_script_on_the_toy_goldens(tmp_path, monkeypatch, golden_dir, capsys):
    """scripts/accept_real_weights.py end to end on a stand-in ProteinGym checkout (the reference's own merge.py, performance script,
    constants and registry, linked, around a four-assay reference file of the toy goldens): run_benchmark writes the CSVs (the device
    served by the reference CLI's golden scores), the reference's scripts turn them into the per-assay Spearman table, and the script
    compares it with a 'published' table -- exit 0 when that table holds the golden CLI scores' Spearman at 3 decimals, 1 when one
    published value is moved by 0.01, 2 (nothing scored, every absent input named) without checkpoints or DMS files."""
    import importlib.util
    import json
    import shutil
    from scipy.stats import spearmanr
    spec = importlib.util.spec_from_file_location("accept_real_weights", os.path.join(os.path.dirname(__file__), "..", "scripts", "accept_real_weights.py"))
    accept = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(accept)
    g = np.load(os.path.join(golden_dir, "golden_esm.npz"), allow_pickle=True)
    _GoldenCliScorer.golden = g
    pg = tmp_path / "ProteinGym"
    (pg / "proteingym").mkdir(parents=True)
    (pg / "reference_files").mkdir()
    for f in ("merge.py", "performance_DMS_benchmarks.py", "constants.json"):
        os.symlink(os.path.join("/root/reference/proteingym", f), pg / "proteingym" / f)
    registry = json.load(open("/root/reference/config.json"))
    registry["model_list_zero_shot_substitutions_DMS"]["ESM2_650M"]["input_score_name"] = "esm2_toy"     # the toy checkpoint's stem
    json.dump(registry, open(pg / "config.json", "w"))
    dms = tmp_path / "dms"
    dms.mkdir()
    layout = [("TOY_A", "TOY_DMS.csv", "seq", "Activity", "medium", "Human"), ("TOY_B", "TOY_DMS.csv", "seq", "Stability", "low", "Virus"),
              ("TOY_C", "TOY_DMS.csv", "seq", "Binding", "high", "Eukaryote"), ("TOY_D", "TOY_DMS.csv", "seq", "Expression", "medium", "Prokaryote")]
    rows = []
    for dms_id, src, seq_key, sel, neff, taxon in layout:
        shutil.copy(os.path.join(golden_dir, src), dms / f"{dms_id}.csv")
        rows.append(dict(DMS_id=dms_id, DMS_filename=f"{dms_id}.csv", target_seq=str(g[seq_key]), DMS_total_number_mutants=len(pd.read_csv(dms / f"{dms_id}.csv")),
                         UniProt_ID=dms_id + "_X", coarse_selection_type=sel, MSA_Neff_L_category=neff, taxon=taxon, MSA_start=1, MSA_end=len(str(g[seq_key]))))
    pd.DataFrame(rows).to_csv(pg / "reference_files" / "DMS_substitutions.csv", index=False)
    clean = json.load(open("/root/reference/proteingym/constants.json"))["clean_names"]
    pub = {}
    for dms_id, src, *_ in layout:
        y = pd.read_csv(dms / f"{dms_id}.csv")["DMS_score"]
        pre = "cli_long/" if "LONG" in src else "cli/"
        one = round(float(spearmanr(y, g[pre + "esm1v_toy_1"])[0]), 3)
        two = np.mean([g[pre + "esm1v_toy_1"], g[pre + "esm1v_toy_2"]], axis=0)
        pub[dms_id] = {clean["ESM1v_single"]: one, clean["ESM1v_ensemble"]: round(float(spearmanr(y, two)[0]), 3),
                       clean["ESM2_650M"]: round(float(spearmanr(y, g[pre + "esm2_toy"])[0]), 3)}
    published = tmp_path / "published.csv"
    pd.DataFrame.from_dict(pub, orient="index").to_csv(published, index_label="DMS ID")

    def fewer_resamples(perf):
        slow = perf.compute_bootstrap_standard_error_functional_categories
        perf.compute_bootstrap_standard_error_functional_categories = lambda df, number_assay_reshuffle=20: slow(df, 20)
    ck = tmp_path / "ckpt"
    ck.mkdir()
    for n in ("esm1v_toy_1.pt", "esm1v_toy_2.pt", "esm2_toy.pt"):
        (ck / n).write_bytes(b"stand-in: the scorer is served by the golden CLI scores")
    argv = ["--proteingym", str(pg), "--dms-folder", str(dms), "--esm1v", str(ck / "esm1v_toy_1.pt"), str(ck / "esm1v_toy_2.pt"), "--esm2", str(ck / "esm2_toy.pt"),
            "--assays", "TOY_A", "TOY_B", "TOY_C", "TOY_D", "--published", str(published)]
    assert accept.main(argv + ["--out", str(tmp_path / "run1")], make_model=_GoldenCliScorer, patch_performance=fewer_resamples) == 0
    report = pd.read_csv(tmp_path / "run1" / "acceptance_report.csv")
    assert len(report) == 12 and (report["verdict"] == "ok").all() and report["abs_difference"].max() <= 0.001 + 1e-12
    assert (np.abs(report["scipy_on_our_scores"] - report["reference_script_on_our_scores"]) <= 0.0005 + 1e-9).all()
    moved = pd.read_csv(published, index_col="DMS ID")
    moved.iloc[1, 0] += 0.01
    moved.to_csv(tmp_path / "moved.csv", index_label="DMS ID")
    argv_moved = argv[:-1] + [str(tmp_path / "moved.csv")]
    assert accept.main(argv_moved + ["--out", str(tmp_path / "run2"), "--assays", "TOY_A", "TOY_B"], make_model=_GoldenCliScorer, patch_performance=fewer_resamples) == 1
    capsys.readouterr()
    assert accept.main(["--proteingym", str(pg), "--dms-folder", str(tmp_path / "no_such_folder"), "--esm1v", str(ck / "absent.pt"), "--assays", "TOY_A", "NOT_AN_ASSAY",
                        "--out", str(tmp_path / "run3")]) == 2
    err = capsys.readouterr().err
    assert "absent.pt" in err and "NOT_AN_ASSAY" in err and "no_such_folder" in err and not os.path.exists(tmp_path / "run3")
    assert accept.main(["--proteingym", str(tmp_path / "nowhere"), "--dms-folder", str(dms), "--esm1v", str(ck / "esm1v_toy_1.pt")]) == 2


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_tranception_slices_vs_live_reference_on_drawn_libraries():
    """tranception.get_sequence_slices and the oracle's restatement against the reference's own function (scoring_utils.py:152-203) on
    drawn inputs: lengths on both sides of a small context, multi-mutant barycentres at the window edges, duplicated rows, the wild type
    among the rows, 'optimal' / 'sliding', indel libraries of mixed lengths -- the same rows in the same order from all three."""
    from hypothesis import given, settings, strategies as st
    from oracle import tranception_oracle as to
    from proteingym_amd import synthetic, tranception as ptr
    rh.load_reference_tranception()
    from tranception.utils import scoring_utils as ref

    @settings(max_examples=120, deadline=None)
    @given(st.integers(0, 2 ** 31 - 1))
    def check(seed):
        rng = np.random.default_rng(seed)
        ctx = int(rng.choice([8, 10, 16, 33]))
        L = int(rng.integers(2, 4 * ctx))
        wt = synthetic.random_sequence(rng, L)
        mode = str(rng.choice(["optimal", "sliding", "indel"]))
        start_idx = int(rng.choice([1, 1, 4]))
        n = int(rng.integers(1, 12))
        if mode == "indel":
            _, seqs = synthetic.random_indel_library(seed=seed % 9973, L=L, n=n, max_edit=min(3, max(1, L - 1)))
            seqs = [s for s in seqs if s] + ([wt] if rng.random() < 0.4 else [])
            df = pd.DataFrame({"mutated_sequence": seqs, "mutant": seqs})
        else:
            muts = []
            for _ in range(n):
                subs = []
                for _ in range(1 if rng.random() < 0.5 else int(rng.integers(2, 5))):
                    p = int(rng.integers(0, L))
                    subs.append(f"{wt[p]}{p + start_idx}{rng.choice([a for a in synthetic.AA if a != wt[p]])}")
                muts.append(":".join(subs))
            if rng.random() < 0.3:
                muts.append(muts[0])
            df = pd.DataFrame({"mutant": muts, "mutated_sequence": [ref.get_mutated_sequence(wt, m, start_idx) for m in muts]})
        kw = dict(start_idx=start_idx, scoring_window="sliding" if mode == "sliding" else "optimal", indel_mode=mode == "indel")
        want = ref.get_sequence_slices(df.copy(), wt, ctx, **kw).reset_index(drop=True)
        cols = ["mutated_sequence", "sliced_mutated_sequence", "window_start", "window_end"]
        for name, fn in (("product", ptr.get_sequence_slices), ("oracle", to.get_sequence_slices)):
            got = fn(df.copy(), wt, ctx, **kw).reset_index(drop=True)
            assert len(got) == len(want) and all(list(got[c]) == list(want[c]) for c in cols), (name, seed, mode, L, ctx)
    check()


@pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present (GPU box)")
def test_tokenizer_vs_live_reference_on_arbitrary_text():
    """The same comparison (esm/data.py:178-254) on text drawn from the characters that matter -- residues in both cases, the rare letters,
    gap symbols, angle brackets and the special tokens' spellings, blanks and newlines: the same ids, the same token lists, or the same
    KeyError."""
    from hypothesis import given, settings, strategies as st
    from proteingym_amd import esm as pesm
    rh.load_reference()
    from esm import data as ref_data
    ref = ref_data.Alphabet.from_architecture("ESM-1b")
    mine = pesm.Alphabet()
    piece = st.one_of(st.sampled_from(["<mask>", "<cls>", "<eos>", "<pad>", "<unk>", "<null_1>", "<", ">", "<m", "mask>", " ", "\n", "\t"]),
                      st.text(alphabet="ACDEFGHIKLMNPQRSTVWYXBUZOJ.-*acdxj", min_size=1, max_size=6))

    @settings(max_examples=500, deadline=None)
    @given(st.lists(piece, min_size=0, max_size=8).map("".join))
    def check(text):
        try:
            want = ref.encode(text)
        except KeyError as e:
            with pytest.raises(KeyError) as got:
                mine.encode(text)
            assert got.value.args == e.args, repr(text)
            return
        assert mine.encode(text) == want, repr(text)
        assert mine.tokenize(text) == ref.tokenize(text), repr(text)
    check()
