"""GPU parity of the Tranception HIP path (through the C ABI) against the reference's own outputs frozen
in tests/golden/golden_tranception.npz (tests/golden/make_golden_tranception.py)."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from proteingym_amd import tranception as ptr

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "golden_tranception.npz"))


@pytest.fixture(scope="module")
def model(lib, golden_dir):
    m = ptr.from_pretrained(os.path.join(golden_dir, "Tranception_toy"))
    yield m
    m.close()


def test_token_logprobs_vs_reference_logits(model, gold):
    ids, mask = gold["logits_ids"], gold["logits_mask"].astype(bool)
    ref = torch.log_softmax(torch.from_numpy(gold["logits"]), -1).numpy()
    lp = model.token_logprobs(ids)
    assert np.abs(lp - ref)[mask].max() < TOL          # rows of [PAD] queries are don't-care


def _merge(df, r):
    return pd.merge(df[["mutated_sequence"]], r, on="mutated_sequence", how="left")


def test_score_mutants_vs_reference(model, gold, golden_dir):
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_DMS.csv"))
    r = model.score_mutants(DMS_data=df, target_seq=str(gold["seq"]), scoring_mirror=True)
    assert list(r.columns) == ["mutated_sequence", "avg_score_L_to_R", "avg_score_R_to_L", "avg_score"]
    r = _merge(df, r)
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(r[c].to_numpy() - gold[f"scores/{c}"]).max() < TOL


def test_long_protein_optimal_window(model, gold, golden_dir):
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_LONG_DMS.csv"))
    r = _merge(df, model.score_mutants(DMS_data=df, target_seq=str(gold["seq_long"]), scoring_mirror=True))
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(r[c].to_numpy() - gold[f"scores_long/{c}"]).max() < TOL


def test_retrieval_vs_reference(lib, gold, golden_dir):
    seq = str(gold["seq"])
    ms, me = [int(v) for v in gold["msa_start_end"]]
    prior = ptr.get_msa_prior(os.path.join(golden_dir, "TOY_MSA.a2m"), None, ms, me, len(seq))
    assert np.abs(prior - gold["msa_prior"]).max() == 0.0
    m = ptr.from_pretrained(os.path.join(golden_dir, "Tranception_toy"),
                            retrieval=dict(MSA_filename=os.path.join(golden_dir, "TOY_MSA.a2m"), MSA_start=ms, MSA_end=me,
                                           full_protein_length=len(seq), retrieval_inference_weight=0.6))
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_DMS.csv"))
    r = _merge(df, m.score_mutants(DMS_data=df, target_seq=seq, scoring_mirror=True))
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(r[c].to_numpy() - gold[f"scores_retrieval/{c}"]).max() < TOL
    m.close()


def test_retrieval_with_eve_weights_vs_reference(lib, gold, golden_dir):
    """--MSA_weights_folder branch: weights file -> MSA_processing -> weighted prior -> fusion on the device."""
    g = np.load(os.path.join(golden_dir, "golden_msa_weights.npz"))
    seq = str(gold["seq"])
    ms, me = [int(v) for v in g["msa_start_end"]]
    m = ptr.from_pretrained(os.path.join(golden_dir, "Tranception_toy"),
                            retrieval=dict(MSA_filename=os.path.join(golden_dir, "TOY_MSA_GAPPY.a2m"), MSA_start=ms, MSA_end=me,
                                           MSA_weight_file_name=os.path.join(golden_dir, "TOY_MSA_GAPPY_weights.npy"),
                                           full_protein_length=len(seq), retrieval_inference_weight=0.6))
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_DMS.csv"))
    r = _merge(df, m.score_mutants(DMS_data=df, target_seq=seq, scoring_mirror=True))
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(r[c].to_numpy() - g[f"scores_retrieval_weighted/{c}"]).max() < TOL
    m.close()


def test_retrieval_on_long_protein_windows_vs_reference(lib, gold, golden_dir):
    """Retrieval on a 1100-residue protein whose alignment covers residues 301..900 only: every scoring window
    overlaps the alignment differently, in both directions (fusion index arithmetic of model_pytorch.py:806-830)."""
    g = np.load(os.path.join(golden_dir, "golden_tranception_long_retrieval.npz"))
    seql = str(gold["seq_long"])
    ms, me = [int(v) for v in g["msa_start_end"]]
    m = ptr.from_pretrained(os.path.join(golden_dir, "Tranception_toy"),
                            retrieval=dict(MSA_filename=os.path.join(golden_dir, "TOY_MSA_LONGSPAN.a2m"), MSA_start=ms, MSA_end=me,
                                           full_protein_length=len(seql), retrieval_inference_weight=0.6))
    dl = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_LONG_DMS.csv"))
    r = pd.merge(dl[["mutated_sequence"]], m.score_mutants(DMS_data=dl, target_seq=seql, scoring_mirror=True), on="mutated_sequence", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(r[c].to_numpy() - g[f"scores/{c}"]).max() < TOL
    m.close()


def test_indel_and_sliding_modes_vs_reference(lib, gold, golden_dir):
    """--indel_mode (variable-length mutated sequences in one padded device batch; the reference's WT row under
    'mutant') and scoring_window='sliding' (per-window log-likelihoods summed per sequence) against the reference."""
    g = np.load(os.path.join(golden_dir, "golden_tranception_modes.npz"))
    seq, seql = str(gold["seq"]), str(gold["seq_long"])
    indel = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_INDEL_DMS.csv"))
    m = ptr.from_pretrained(os.path.join(golden_dir, "Tranception_toy"))
    r = m.score_mutants(DMS_data=indel, target_seq=seq, scoring_mirror=True, indel_mode=True)
    assert sorted(r.columns) == sorted(g["indel/columns"])
    wt = r[r["mutated_sequence"].isna()]
    assert len(wt) == 1 and wt["mutant"].iloc[0] == seq and float(wt["avg_score"].iloc[0]) == 0.0
    rr = pd.merge(indel[["mutated_sequence"]].iloc[1:], r, on="mutated_sequence", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(rr[c].to_numpy() - g[f"indel/{c}"]).max() < TOL
    m.close()
    ms = ptr.from_pretrained(os.path.join(golden_dir, "Tranception_toy"), scoring_window="sliding")
    dl = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_LONG_DMS.csv"))
    rs = pd.merge(dl[["mutated_sequence"]], ms.score_mutants(DMS_data=dl, target_seq=seql, scoring_mirror=True), on="mutated_sequence", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        assert np.abs(rs[c].to_numpy() - g[f"sliding/{c}"]).max() < TOL
    ms.close()


def test_wt_row_and_determinism(model, gold, golden_dir):
    seq = str(gold["seq"])
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_DMS.csv")).iloc[:5]
    df = pd.concat([df, pd.DataFrame({"mutant": ["WT"], "mutated_sequence": [seq]})], ignore_index=True)
    df.loc[df.index[-1], "mutant"] = df["mutant"].iloc[0]      # a syntactically valid triplet for the WT row's slice
    a = model.score_mutants(DMS_data=df, target_seq=seq)
    b = model.score_mutants(DMS_data=df, target_seq=seq)
    assert a.equals(b)
    wt = a[a.mutated_sequence == seq]
    assert len(wt) == 1 and float(wt["avg_score"].iloc[0]) == 0.0


def test_cli_writes_reference_csv(lib, gold, golden_dir, tmp_path):
    from proteingym_amd import score_tranception_proteingym as cli
    args = cli.create_parser().parse_args([
        "--checkpoint", os.path.join(golden_dir, "Tranception_toy"), "--target_seq", str(gold["seq"]),
        "--DMS_file_name", "TOY_TRANCEPTION_DMS.csv", "--DMS_data_folder", golden_dir,
        "--output_scores_folder", str(tmp_path / "out"), "--inference_time_retrieval", "--MSA_folder", golden_dir,
        "--MSA_filename", "TOY_MSA.a2m", "--MSA_start", str(int(gold["msa_start_end"][0]) + 1),
        "--MSA_end", str(int(gold["msa_start_end"][1]))])
    cli.main(args)
    out = pd.read_csv(tmp_path / "out" / "TOY_TRANCEPTION_DMS.csv")
    assert list(out.columns) == ["mutated_sequence", "avg_score_L_to_R", "avg_score_R_to_L", "avg_score"]
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_DMS.csv"))
    r = _merge(df, out)
    assert np.abs(r["avg_score"].to_numpy() - gold["scores_retrieval/avg_score"]).max() < TOL


def test_run_sharded_mutant_chunks_equal_the_cli(lib, gold, golden_dir, tmp_path):
    """run_sharded tranception (config 4's multi-GPU runner) on one GPU with the work cut into chunks of <= 7 rows: one
    checkpoint load, the retrieval prior swapped in per assay (assay 0 with retrieval arguments from the reference table,
    assay 1 = the same file under another id), per-row scores gathered and re-assembled -- the CSVs must be byte-identical
    to the single-assay CLI's (a sequence's bits do not depend on what shares its batch)."""
    from proteingym_amd import run_sharded, score_tranception_proteingym as cli
    src = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_DMS.csv"))
    dms = tmp_path / "dms"
    dms.mkdir()
    src.to_csv(dms / "A.csv", index=False)
    src.iloc[::-1].to_csv(dms / "B.csv", index=False)
    s0, s1 = int(gold["msa_start_end"][0]) + 1, int(gold["msa_start_end"][1])
    ref = pd.DataFrame({"DMS_id": ["A", "B"], "DMS_filename": ["A.csv", "B.csv"], "target_seq": [str(gold["seq"])] * 2,
                        "DMS_total_number_mutants": [len(src)] * 2, "MSA_filename": ["TOY_MSA.a2m"] * 2,
                        "MSA_start": [s0] * 2, "MSA_end": [s1] * 2, "weight_file_name": ["none.npy"] * 2})
    ref.to_csv(tmp_path / "ref.csv", index=False)
    common = ["--checkpoint", os.path.join(golden_dir, "Tranception_toy"), "--DMS_reference_file_path", str(tmp_path / "ref.csv"),
              "--DMS_data_folder", str(dms), "--inference_time_retrieval", "--MSA_folder", golden_dir]
    items = run_sharded.main(["tranception", "--max-chunk-rows", "7", "--", *common, "--output_scores_folder", str(tmp_path / "sharded")])
    assert len(items) == 2 * -(-len(src) // 7)
    for i, name in enumerate(("A", "B")):
        cli.main(cli.create_parser().parse_args(common + ["--output_scores_folder", str(tmp_path / "single"), "--DMS_index", str(i)]))
        assert open(tmp_path / "sharded" / f"{name}.csv").read() == open(tmp_path / "single" / f"{name}.csv").read()


def test_token_logprobs_do_not_depend_on_the_batch_or_on_the_gemm_item_kind(lib, golden_dir, gemm_option):
    """The first sequences of a batch of 7 / 100 / 200: the small launches run every GEMM tile as two half-height items, the
    200-sequence one as full-height items (2 x 228 tiles do not fit one round of CUs), the option gemm_half_tail = 0 forces full-height
    items at every size -- the same bits in all of them, through the split-plane (c_fc, squared ReLU) and the fp32 epilogues.
    (Round 4: a rewrite of the split-plane epilogue passed every op-level comparison on random data and differed here, on rows
    holding values below fp16's normal range.)"""
    m = ptr.from_pretrained(os.path.join(golden_dir, "Tranception_toy"), device=0)
    rng = np.random.default_rng(0)
    seqs = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), size=70)) for _ in range(200)]
    ids, _ = m.encode_batch(seqs)
    gemm_option("gemm_half_tail", 1)
    base = m.token_logprobs(ids[:7])
    for half in (0, 1):
        gemm_option("gemm_half_tail", half)
        for B in (7, 100, 200):
            assert np.array_equal(m.token_logprobs(ids[:B])[:7], base), (half, B)
    m.close()


def test_context_edges_around_1022_residues_vs_oracle(lib):
    """The Tranception context holds 1 022 residues + [CLS] / [SEP] (scoring_utils.py:152-203: a protein that fits is scored
    whole, a longer one through the optimal window of every mutant, Delta to the wild type OF THE SAME WINDOW): proteins of
    1 021 ... 1 025 residues on both sides of that edge, mutants at the ends, at the window's centre and just off it, both
    reading directions, against the oracle's scorer at a narrow width (2 layers x 256, 4 heads of 64)."""
    import torch
    from oracle import tranception_oracle as to
    from proteingym_amd import synthetic, tranception as ptr
    cfg = dict(synthetic.TRANCEPTION_L, layers=2, embed_dim=256, heads=4, ffn_dim=1024)
    blob = synthetic.random_tranception_weights(cfg, seed=5)
    model = ptr.TranceptionModel(cfg, blob, device=0)
    ocfg, W = to.from_arrays(arrays=synthetic.tranception_blob_to_arrays(cfg, blob), **cfg)
    rng = np.random.default_rng(77)
    for L in (1021, 1022, 1023, 1025):
        wt = "".join(rng.choice(list(synthetic.AA), size=L))
        muts = [f"{wt[p - 1]}{p}{'A' if wt[p - 1] != 'A' else 'C'}" for p in (1, 2, 510, 511, 512, 513, L // 2, L - 511, L - 1, L)]
        muts.append(":".join(muts[::4]))
        df = pd.DataFrame({"mutant": muts, "mutated_sequence": [ptr.get_mutated_sequence(wt, m) for m in muts]})
        with torch.no_grad():
            want = to.score_mutants(ocfg, W, df, wt)
        have = model.score_mutants(DMS_data=df, target_seq=wt)
        assert list(have["mutated_sequence"]) == list(want["mutated_sequence"])
        worst = max(float(np.abs(have[c].to_numpy() - want[c].to_numpy()).max()) for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"))
        print(f"Tranception, {L} residues: avg scores max|err| {worst:.2e}")
        assert worst < TOL
    model.close()


def test_indels_with_retrieval_vs_reference(lib, golden_dir, tmp_path):
    """Indel scoring WITH inference-time retrieval through the CLI (score_tranception_proteingym --indel_mode --inference_time_retrieval
    --clustal_omega_location ...): every sequence re-aligned (here by tests/golden/stand_in_clustalo.py, the aligner the unmodified
    reference was given when tests/golden/make_golden_tranception_indel_retrieval.py froze its columns), the prior re-indexed per
    sequence, inserted residues left to the network; an alignment that does not span the protein fails like the reference (IndexError)."""
    import shutil
    import stat
    from proteingym_amd import score_tranception_proteingym as cli
    g = np.load(os.path.join(golden_dir, "golden_tranception_indel_retrieval.npz"))
    seq = str(g["seq"])
    aligner = tmp_path / "clustalo"                                     # an executable of our own: mode bits need not survive the trip to the box
    aligner.write_text('#!/bin/sh\nexec python3 "%s" "$@"\n' % os.path.join(golden_dir, "stand_in_clustalo.py"))
    aligner.chmod(aligner.stat().st_mode | stat.S_IXUSR)
    msa = tmp_path / "msa"
    msa.mkdir()
    shutil.copy(os.path.join(golden_dir, "TOY_MSA_INDEL_FULL.a2m"), msa / "TOY_MSA_INDEL_FULL.a2m")
    shutil.copy(os.path.join(golden_dir, "TOY_MSA.a2m"), msa / "TOY_MSA.a2m")
    df = pd.read_csv(os.path.join(golden_dir, "TOY_TRANCEPTION_INDEL_RETRIEVAL_DMS.csv"))
    common = ["--checkpoint", os.path.join(golden_dir, "Tranception_toy"), "--DMS_data_folder", golden_dir,
              "--DMS_file_name", "TOY_TRANCEPTION_INDEL_RETRIEVAL_DMS.csv", "--target_seq", seq, "--indel_mode", "--inference_time_retrieval",
              "--MSA_folder", str(msa), "--clustal_omega_location", str(aligner), "--batch_size_inference", "1"]
    cli.main(cli.create_parser().parse_args(common + ["--MSA_filename", "TOY_MSA_INDEL_FULL.a2m", "--MSA_start", "1", "--MSA_end", str(len(seq)),
                                                      "--output_scores_folder", str(tmp_path / "out")]))
    r = pd.read_csv(tmp_path / "out" / "TOY_TRANCEPTION_INDEL_RETRIEVAL_DMS.csv", float_precision="round_trip")
    key = r["mutated_sequence"].fillna(r["mutant"]) if "mutant" in r else r["mutated_sequence"]      # the wild type's zero row sits under 'mutant'
    m = pd.merge(df[["mutated_sequence"]], r.assign(key=key), left_on="mutated_sequence", right_on="key", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        err = np.abs(m[c].to_numpy(dtype=np.float64) - g[f"full/{c}"]).max()
        print(f"indels with retrieval, {c}: max|err| {err:.2e}")
        assert err < TOL, c
    with pytest.raises(IndexError):
        cli.main(cli.create_parser().parse_args(common + ["--MSA_filename", "TOY_MSA.a2m", "--MSA_start", "11", "--MSA_end", "60",
                                                          "--output_scores_folder", str(tmp_path / "out2")]))


# ---- prefix-shared scoring (pgmi_tr_sequence_loglik_shared): the bits of the full forward ---------------------------------
def _shared_vs_full(model, wt, mutants, reverse=False, retrieval=None, token_level=True):
    """out / token log-probs of the shared entry against the unshared entries on [wild type] + mutants (all of one length)."""
    import ctypes as C
    from proteingym_amd import _lib
    lib = _lib.load()
    seqs = [wt] + list(mutants)
    if reverse:
        seqs = [s[::-1] for s in seqs]
    ids, lens = model.encode_batch(seqs)
    B, T = ids.shape
    ref = np.zeros(B, dtype=np.int32)
    prior = (None, 0, None, None, None, None, 0.0)
    if retrieval is not None:
        lp = _lib.as_f32(retrieval["log_prior"])
        a0 = np.full(B, retrieval["a0"], np.int32); r0 = np.full(B, retrieval["row0"], np.int32)
        nn = np.full(B, retrieval["n"], np.int32); fl = np.full(B, 1 if reverse else 0, np.int32)
        prior = (_lib.ptr(lp, _lib._f32p), lp.shape[0], _lib.ptr(a0, _lib._i32p), _lib.ptr(r0, _lib._i32p), _lib.ptr(nn, _lib._i32p),
                 _lib.ptr(fl, _lib._i32p), 0.6)
    full = np.empty(B, np.float32)
    _lib.check(lib.pgmi_tr_sequence_loglik(model._h, _lib.ptr(ids, _lib._i32p), _lib.ptr(lens, _lib._i32p), B, T, *prior, _lib.ptr(full, _lib._f32p)))
    shared = np.empty(B, np.float32)
    tok = np.empty((B, T, 25), np.float32) if token_level else None
    rows = np.zeros(1, np.int64)
    _lib.check(lib.pgmi_tr_sequence_loglik_shared(model._h, _lib.ptr(ids, _lib._i32p), _lib.ptr(ref, _lib._i32p), B, T, *prior,
                                                  _lib.ptr(shared, _lib._f32p), _lib.ptr(tok, _lib._f32p) if token_level else None,
                                                  _lib.ptr(rows, _lib._i64p)))
    assert np.array_equal(shared, full), np.abs(shared - full).max()
    if token_level:
        assert np.array_equal(tok, model.token_logprobs(ids))
    return int(rows[0]), B * T


def _mutants_everywhere(wt, rng, n_multi=12):
    aa = "ACDEFGHIKLMNPQRSTVWY"
    L = len(wt)

    def sub(s, p):
        return s[:p] + rng.choice([c for c in aa if c != s[p]]) + s[p + 1:]
    out = [sub(wt, p) for p in list(range(0, L, 3)) + [L - 1, L - 2, 30, 31, 32, 63, 64]if p < L]
    for _ in range(n_multi):
        s = wt
        for p in rng.choice(L, size=int(rng.integers(2, 6)), replace=False):
            s = sub(s, int(p))
        out.append(s)
    out.append(wt)                                                    # a copy of the root
    return out


@pytest.mark.parametrize("L", [12, 30, 40, 62, 94, 126, 200])
def test_prefix_shared_scoring_has_the_bits_of_the_full_forward(model, L):
    """Sequence log-likelihoods and every token log-prob row of the shared entry == the unshared entries: mutants at every third
    position (first residue, last residue, both sides of the 32-token tile edges), multi-mutants, a copy of the wild type; T = L + 2 on
    and off multiples of 32; both reading directions; with a retrieval prior fused."""
    rng = np.random.default_rng(L)
    wt = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), size=L))
    muts = _mutants_everywhere(wt, rng)
    done, full = _shared_vs_full(model, wt, muts)
    assert done < full and (L < 64 or done < 0.7 * full)
    _shared_vs_full(model, wt, muts, reverse=True)
    prior = np.log(rng.dirichlet(np.ones(25), size=L + 10)).astype(np.float32)
    _shared_vs_full(model, wt, muts, retrieval=dict(log_prior=prior, a0=3, row0=5, n=L - 8), token_level=False)
    _shared_vs_full(model, wt, muts, reverse=True, retrieval=dict(log_prior=prior, a0=2, row0=4, n=L - 8), token_level=False)


def test_prefix_shared_scoring_one_two_and_three_query_tiles(model):
    """T = 14 ... 96 tokens: one, two and three 32-token tiles per sequence, i.e. the one-, two- and four-wave instantiations of the attention
    kernel (the ragged launch uses the instantiation the dense launch of that T uses), T exactly 32 and 64 included."""
    rng = np.random.default_rng(3)
    for L in (12, 30, 31, 62, 63, 94):
        wt = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), size=L))
        _shared_vs_full(model, wt, _mutants_everywhere(wt, rng, n_multi=3))
        _shared_vs_full(model, wt, _mutants_everywhere(wt, rng, n_multi=3), reverse=True)


def test_prefix_shared_scoring_in_chunks_and_groups(lib, golden_dir):
    """A workspace of 2 048 rows: the groups are cut into chunks, every chunk carries its root again; several roots (windows) in one
    call, a root without members, members listed before their root."""
    import ctypes as C
    from proteingym_amd import _lib
    m = ptr.from_pretrained(os.path.join(golden_dir, "Tranception_toy"), max_rows=2048)
    rng = np.random.default_rng(7)
    L = 150
    roots = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), size=L)) for _ in range(3)]
    seqs, ref = [], []
    for k, wt in enumerate(roots[:2]):
        mut = _mutants_everywhere(wt, rng, n_multi=4)
        base = len(seqs)
        seqs += mut[:5] + [wt] + mut[5:]                              # the root in the middle of its group
        ref += [base + 5] * (len(mut) + 1)
    seqs.append(roots[2]); ref.append(len(seqs) - 1)                  # a root on its own
    ids, lens = m.encode_batch(seqs)
    B, T = ids.shape
    full = np.empty(B, np.float32)
    _lib.check(_lib.load().pgmi_tr_sequence_loglik(m._h, _lib.ptr(ids, _lib._i32p), _lib.ptr(lens, _lib._i32p), B, T,
                                                   None, 0, None, None, None, None, 0.0, _lib.ptr(full, _lib._f32p)))
    shared = np.empty(B, np.float32)
    rows = np.zeros(1, np.int64)
    r = np.asarray(ref, dtype=np.int32)
    _lib.check(_lib.load().pgmi_tr_sequence_loglik_shared(m._h, _lib.ptr(ids, _lib._i32p), _lib.ptr(r, _lib._i32p), B, T, None, 0, None, None,
                                                          None, None, 0.0, _lib.ptr(shared, _lib._f32p), None, _lib.ptr(rows, _lib._i64p)))
    assert np.array_equal(shared, full)
    assert rows[0] > 2048                                             # more than one chunk ran
    bad = r.copy(); bad[0] = 1                                        # sequence 1 is not a root
    with pytest.raises(_lib.PgmiError, match="not a root"):
        _lib.check(_lib.load().pgmi_tr_sequence_loglik_shared(m._h, _lib.ptr(ids, _lib._i32p), _lib.ptr(bad, _lib._i32p), B, T, None, 0, None,
                                                              None, None, None, 0.0, _lib.ptr(shared, _lib._f32p), None, None))
    m.close()


def test_score_mutants_is_the_same_frame_with_and_without_prefix_sharing(model, gold, golden_dir):
    """TranceptionModel.score_mutants (slices, wild-type delta, mirror average) with share_prefix on and off: the same bits in every
    column, for the short protein, the 1 100-residue protein (a window per mutated position) and the 'sliding' window."""
    for csv, key in (("TOY_TRANCEPTION_DMS.csv", "seq"), ("TOY_TRANCEPTION_LONG_DMS.csv", "seq_long")):
        df = pd.read_csv(os.path.join(golden_dir, csv))
        for window in ("optimal", "sliding"):
            model.scoring_window = window
            try:
                model.share_prefix = True
                model.rows_forwarded = model.rows_full = 0
                a = model.score_mutants(DMS_data=df, target_seq=str(gold[key]), scoring_mirror=True)
                saved = (model.rows_forwarded, model.rows_full)
                model.share_prefix = False
                b = model.score_mutants(DMS_data=df, target_seq=str(gold[key]), scoring_mirror=True)
            finally:
                model.share_prefix, model.scoring_window = True, "optimal"
            assert list(a["mutated_sequence"]) == list(b["mutated_sequence"])
            for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
                assert np.array_equal(a[c].to_numpy(), b[c].to_numpy()), (csv, window, c)
            assert saved[0] < saved[1]


def test_prefix_shared_scoring_at_the_large_width(lib):
    """Tranception-L's layer shape (1280 wide, 20 heads: five ALiBi slopes, FFN 5120; 4 layers), a 286-residue protein and a
    1 022-residue one (32 key tiles): shared == full, bit for bit, in both directions; the rows forwarded are counted."""
    from proteingym_amd import synthetic
    cfg = dict(synthetic.TRANCEPTION_L, layers=4)
    m = ptr.TranceptionModel(cfg, synthetic.random_tranception_weights(cfg, seed=5), device=0)
    rng = np.random.default_rng(1)
    for L, n in ((286, 60), (1022, 12)):
        wt = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), size=L))
        muts = []
        for p in rng.choice(L, size=n, replace=False):
            p = int(p)
            muts.append(wt[:p] + ("A" if wt[p] != "A" else "C") + wt[p + 1:])
        done, full = _shared_vs_full(m, wt, muts, token_level=False)
        done_r, _ = _shared_vs_full(m, wt, muts, reverse=True, token_level=False)
        print(f"L = {L}: rows forwarded {done} + {done_r} of 2 x {full}")
        assert done + done_r < 1.15 * full                             # uniform positions: ~half of the rows per direction, + the root
    m.close()


def test_multi_mutants_share_intermediate_roots_same_bits(model):
    """A pairwise library (3 x 4 first substitutions, each with 5 x 3 second ones) and random triples: with intermediate roots ("wild type +
    first substitution", tranception.intermediate_roots) the same frame as with the wild type as the only root and as with every sequence in
    full -- bit for bit, both directions -- and fewer rows forwarded."""
    rng = np.random.default_rng(11)
    L = 120
    aa = "ACDEFGHIKLMNPQRSTVWY"
    wt = "".join(rng.choice(list(aa), size=L))

    def name(p, c):
        c = c if wt[p] != c else ("A" if c != "A" else "C")
        return f"{wt[p]}{p + 1}{c}"
    muts = []
    for i in (5, 40, 70):
        for a in "DEKR":
            for j in (80, 90, 100, 110, 118):
                for b in "GPW":
                    muts.append(name(i, a) + ":" + name(j, b))
    for _ in range(40):
        pos = sorted(int(p) for p in rng.choice(L, size=3, replace=False))
        muts.append(":".join(name(p, str(rng.choice(list(aa)))) for p in pos))
    muts.append(name(5, "D"))
    muts = list(dict.fromkeys(muts))
    df = pd.DataFrame({"mutant": muts, "mutated_sequence": ptr.mutated_sequences(wt, muts)}).drop_duplicates("mutated_sequence")
    frames, rows = {}, {}
    for name, (share, inter) in {"full": (False, False), "wild type": (True, False), "intermediate": (True, True)}.items():
        model.share_prefix, model.share_intermediate = share, inter
        model.rows_forwarded = model.rows_full = 0
        try:
            frames[name] = model.score_mutants(DMS_data=df, target_seq=wt, scoring_mirror=True)
        finally:
            model.share_prefix, model.share_intermediate = True, True
        rows[name] = model.rows_forwarded
    for name in ("wild type", "intermediate"):
        assert list(frames[name]["mutated_sequence"]) == list(frames["full"]["mutated_sequence"])
        for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
            assert np.array_equal(frames[name][c].to_numpy(), frames["full"][c].to_numpy()), (name, c)
    print("rows forwarded:", rows)
    assert rows["intermediate"] < 0.8 * rows["wild type"] < 0.8 * rows["full"]
