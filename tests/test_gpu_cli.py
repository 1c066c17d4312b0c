"""The drop-in CLI (proteingym_amd/compute_fitness.py) and the multi-assay runner on a GPU:
output CSVs equal the reference CLI's (columns frozen in tests/golden/golden_esm.npz)."""
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _run_cli(argv):
    from proteingym_amd import compute_fitness as cf
    cf.main(cf.create_parser().parse_args(argv))


def test_cli_masked_marginals_ensemble(lib, golden, golden_dir, tmp_path):
    out = tmp_path / "o"
    _run_cli(["--model-location", os.path.join(golden_dir, "esm1v_toy_1.pt"), os.path.join(golden_dir, "esm1v_toy_2.pt"),
              "--model_type", "ESM1v", "--dms-input", os.path.join(golden_dir, "TOY_DMS.csv"), "--dms-output", str(out),
              "--target_seq", str(golden["seq"]), "--scoring-strategy", "masked-marginals", "--scoring-window", "optimal"])
    df = pd.read_csv(out / "TOY_DMS.csv")
    assert list(df.columns) == list(golden["cli/columns"])      # same columns, same order
    for c in ("esm1v_toy_1", "esm1v_toy_2", "Ensemble_ESM1v"):
        assert np.abs(df[c].to_numpy() - golden[f"cli/{c}"]).max() < TOL
    src = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv"))
    assert df[list(src.columns)].equals(src)                     # input columns preserved


def test_cli_esm2_has_no_ensemble_column(lib, golden, golden_dir, tmp_path):
    out = tmp_path / "o"
    _run_cli(["--model-location", os.path.join(golden_dir, "esm2_toy.pt"), "--model_type", "ESM2",
              "--dms-input", os.path.join(golden_dir, "TOY_DMS.csv"), "--dms-output", str(out),
              "--target_seq", str(golden["seq"]), "--scoring-strategy", "masked-marginals"])
    df = pd.read_csv(out / "TOY_DMS.csv")
    assert "Ensemble_ESM1v" not in df.columns
    assert np.abs(df["esm2_toy"].to_numpy() - golden["cli/esm2_toy"]).max() < TOL


def test_cli_wt_marginals_short_and_overlapping(lib, golden, golden_dir, tmp_path):
    out = tmp_path / "o"
    _run_cli(["--model-location", os.path.join(golden_dir, "esm1b_toy_lnb.pt"), "--model_type", "ESM1b",
              "--dms-input", os.path.join(golden_dir, "TOY_DMS.csv"), "--dms-output", str(out),
              "--target_seq", str(golden["seq"]), "--scoring-strategy", "wt-marginals"])
    df = pd.read_csv(out / "TOY_DMS.csv")
    assert np.abs(df["esm1b_toy_lnb"].to_numpy() - golden["cli_wt/esm1b_toy_lnb"]).max() < TOL
    out2 = tmp_path / "o2"
    _run_cli(["--model-location", os.path.join(golden_dir, "esm1v_toy_1.pt"), "--model_type", "ESM1b",
              "--dms-input", os.path.join(golden_dir, "TOY_LONG_DMS.csv"), "--dms-output", str(out2),
              "--target_seq", str(golden["seq_long"]), "--scoring-strategy", "wt-marginals",
              "--scoring-window", "overlapping"])
    df = pd.read_csv(out2 / "TOY_LONG_DMS.csv")
    assert np.abs(df["esm1v_toy_1"].to_numpy() - golden["cli_wt_long/esm1v_toy_1"]).max() < TOL


def test_cli_pseudo_ppl(lib, golden, golden_dir, tmp_path):
    src = pd.read_csv(os.path.join(golden_dir, "TOY_DMS.csv")).iloc[:6][["mutant", "DMS_score"]]
    src.to_csv(tmp_path / "TOY_PPPL.csv", index=False)
    out = tmp_path / "o"
    _run_cli(["--model-location", os.path.join(golden_dir, "esm2_toy.pt"), "--model_type", "ESM2",
              "--dms-input", str(tmp_path / "TOY_PPPL.csv"), "--dms-output", str(out),
              "--target_seq", str(golden["seq"]), "--scoring-strategy", "pseudo-ppl"])
    df = pd.read_csv(out / "TOY_PPPL.csv")
    assert "mutated_sequence" in df.columns
    assert np.abs(df["esm2_toy"].to_numpy() - golden["cli_pppl/esm2_toy"]).max() < 1e-4 * np.sqrt(68)   # sums of 68 terms, each term held to 1e-4 (tests/test_gpu_pppl.py)


def test_cli_dms_index_mapping_and_runner(lib, golden, golden_dir, tmp_path):
    """--dms_index/--dms_mapping resolution (compute_fitness.py:288-305) and the multi-assay runner."""
    mapping = pd.DataFrame({"DMS_id": ["TOY_A", "TOY_B"], "DMS_filename": ["TOY_DMS.csv", "TOY_DMS.csv"],
                            "target_seq": [str(golden["seq"]).lower(), str(golden["seq"])],
                            "DMS_total_number_mutants": [100, 100]})
    mapping.to_csv(tmp_path / "map.csv", index=False)
    out = tmp_path / "o"
    _run_cli(["--model-location", os.path.join(golden_dir, "esm1v_toy_1.pt"), "--model_type", "ESM1v",
              "--dms_index", "0", "--dms_mapping", str(tmp_path / "map.csv"), "--dms-input", golden_dir,
              "--dms-output", str(out), "--scoring-strategy", "masked-marginals"])
    df = pd.read_csv(out / "TOY_A.csv")
    assert np.abs(df["esm1v_toy_1"].to_numpy() - golden["cli/esm1v_toy_1"]).max() < TOL
    assert np.abs(df["Ensemble_ESM1v"].to_numpy() - golden["cli/esm1v_toy_1"]).max() < TOL
    from proteingym_amd import run_benchmark as rb
    out2 = tmp_path / "o2"
    rb.main(rb.create_parser().parse_args(
        ["--model-location", os.path.join(golden_dir, "esm1v_toy_1.pt"), os.path.join(golden_dir, "esm1v_toy_2.pt"),
         "--model_type", "ESM1v", "--dms_mapping", str(tmp_path / "map.csv"), "--dms-input", golden_dir,
         "--dms-output", str(out2)]))
    for name in ("TOY_A", "TOY_B"):
        df = pd.read_csv(out2 / f"{name}.csv")
        assert list(df.columns) == list(golden["cli/columns"])
        for c in ("esm1v_toy_1", "esm1v_toy_2", "Ensemble_ESM1v"):
            assert np.abs(df[c].to_numpy() - golden[f"cli/{c}"]).max() < TOL


def test_runner_position_shards_equal_assay_shards(lib, golden, golden_dir, tmp_path):
    """run_benchmark --shard positions (tables assembled from chunks of masked positions, scored on the host) writes
    the same numbers as the default --shard assay (device scoring), bit for bit, incl. a 1100-residue protein whose
    chunks use different 1024-token windows."""
    from proteingym_amd import run_benchmark as rb
    mapping = pd.DataFrame({"DMS_id": ["TOY_A", "TOY_LONG"], "DMS_filename": ["TOY_DMS.csv", "TOY_LONG_DMS.csv"],
                            "target_seq": [str(golden["seq"]), str(golden["seq_long"])]})
    mapping.to_csv(tmp_path / "map.csv", index=False)
    common = ["--model-location", os.path.join(golden_dir, "esm1v_toy_1.pt"), os.path.join(golden_dir, "esm1v_toy_2.pt"),
              "--model_type", "ESM1v", "--dms_mapping", str(tmp_path / "map.csv"), "--dms-input", golden_dir]
    rb.main(rb.create_parser().parse_args(common + ["--dms-output", str(tmp_path / "a")]))
    rb.main(rb.create_parser().parse_args(common + ["--dms-output", str(tmp_path / "p"), "--shard", "positions",
                                                    "--chunk-forwards", "7"]))
    for name in ("TOY_A", "TOY_LONG"):
        a = pd.read_csv(tmp_path / "a" / f"{name}.csv", float_precision="round_trip")
        p = pd.read_csv(tmp_path / "p" / f"{name}.csv", float_precision="round_trip")
        assert list(a.columns) == list(p.columns)
        for c in ("esm1v_toy_1", "esm1v_toy_2", "Ensemble_ESM1v"):
            assert np.array_equal(a[c].to_numpy(), p[c].to_numpy())
    assert np.abs(p["esm1v_toy_1"].to_numpy() - golden["cli_long/esm1v_toy_1"]).max() < TOL


def test_runner_wt_marginals_on_a_clinical_shaped_mapping_equals_the_cli(lib, golden, golden_dir, tmp_path):
    """The fifth caller of the ESM scorer (scripts/scoring_clinical_zero_shot/scoring_ESM1b_substitutions.sh:29-37): wt-marginals
    with overlapping windows over a mapping with the CLINICAL column set (file_length, no DMS_total_number_mutants, no
    start_idx / DMS_mutant_column).  run_benchmark keeps the checkpoint resident across genes; its CSVs must hold the same
    numbers, bit for bit, and the same columns as one single-assay CLI run per --dms_index (a gene > 1022 residues included)."""
    from proteingym_amd import run_benchmark as rb
    genes = [("NP_TOY_A.1", "TOY_DMS.csv", str(golden["seq"])), ("NP_TOY_LONG.2", "TOY_LONG_DMS.csv", str(golden["seq_long"])),
             ("NP_TOY_B.1", "TOY_DMS.csv", str(golden["seq"]))]
    mapping = pd.DataFrame({"DMS_id": [g[0] for g in genes], "target_seq": [g[2] for g in genes], "file_length": [len(g[2]) for g in genes],
                            "DMS_filename": [g[1] for g in genes], "EVE_model_path": ["x"] * 3, "MSA_filename": ["x.a2m"] * 3,
                            "alignment_source": ["Invitae"] * 3, "weight_file_name": ["x.npy"] * 3, "MSA_start": [1] * 3,
                            "MSA_end": [len(g[2]) for g in genes], "MSA_len": [len(g[2]) for g in genes]})
    mapping.to_csv(tmp_path / "clinical.csv", index=False)
    ck = os.path.join(golden_dir, "esm1v_toy_1.pt")
    rb.main(rb.create_parser().parse_args(["--model-location", ck, "--model_type", "ESM1b", "--dms_mapping", str(tmp_path / "clinical.csv"),
                                           "--dms-input", golden_dir, "--dms-output", str(tmp_path / "runner"),
                                           "--scoring-strategy", "wt-marginals", "--scoring-window", "overlapping"]))
    for i, (gene, _, _) in enumerate(genes):
        _run_cli(["--model-location", ck, "--model_type", "ESM1b", "--dms-input", golden_dir, "--dms-output", str(tmp_path / "cli"),
                  "--scoring-strategy", "wt-marginals", "--scoring-window", "overlapping", "--dms_mapping", str(tmp_path / "clinical.csv"),
                  "--dms_index", str(i)])
        a = pd.read_csv(tmp_path / "cli" / f"{gene}.csv", float_precision="round_trip")
        b = pd.read_csv(tmp_path / "runner" / f"{gene}.csv", float_precision="round_trip")
        assert list(a.columns) == list(b.columns) and "Ensemble_ESM1v" not in b.columns
        assert np.array_equal(a["esm1v_toy_1"].to_numpy(), b["esm1v_toy_1"].to_numpy())
    assert np.abs(b["esm1v_toy_1"].to_numpy() - golden["cli/esm1v_toy_1"]).max() > 1e-3        # a different strategy, not a relabelled masked-marginals column
    long = pd.read_csv(tmp_path / "runner" / "NP_TOY_LONG.2.csv")
    assert np.abs(long["esm1v_toy_1"].to_numpy() - golden["cli_wt_long/esm1v_toy_1"]).max() < TOL   # = the reference CLI's overlapping-window column
    assert [rb.wt_marginals_windows(n, "overlapping") for n in (500, 1024, 1025, 1535, 2047, 3425)] == [1, 1, 2, 2, 4, 6]


@pytest.mark.parametrize("ckpts,model_type,extra", [(("esm1v_toy_1.pt", "esm1v_toy_2.pt"), "ESM1v", []), (("esm2_toy.pt",), "ESM2", []),
                                                    (("esm1b_toy_lnb.pt",), "ESM1b", ["--all-positions"])])
def test_runner_short_assay_groups_write_the_one_at_a_time_files(lib, golden_dir, tmp_path, ckpts, model_type, extra):
    """run_benchmark scores short assays several at a time (every masked copy of every member in one launch sequence, padded to
    the longest member behind a key mask, scored on the host from the table rows).  The CSVs must be byte-identical to the ones
    written one assay at a time (--batch-short-tokens 0 = Assay.run per assay): lengths on both sides of the 32-key tile edges,
    learned positions + token dropout (ESM-1v), LN-before (ESM-1b), rotary (ESM2), multi-mutants."""
    from proteingym_amd import run_benchmark as rb, synthetic
    rows = []
    for k, L in enumerate((31, 40, 47, 62, 63, 90, 97, 130, 30)):
        seq, muts, score = synthetic.random_assay(seed=40 + k, L=L, n_single=3 * L // 2, n_multi=12)
        pd.DataFrame({"mutant": muts, "DMS_score": score}).to_csv(tmp_path / f"G{k}.csv", index=False)
        rows.append({"DMS_id": f"G{k}", "DMS_filename": f"G{k}.csv", "target_seq": seq, "DMS_total_number_mutants": len(muts)})
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    common = ["--model-location", *[os.path.join(golden_dir, c) for c in ckpts], "--model_type", model_type,
              "--dms_mapping", str(tmp_path / "map.csv"), "--dms-input", str(tmp_path), *extra]
    st = rb.main(rb.create_parser().parse_args(common + ["--dms-output", str(tmp_path / "grouped"), "--batch-group-rows", "20000"]))
    sizes = sorted({e.get("group_of", 1) for e in st["rank0_assays"]})
    assert sizes[-1] > 1 and len([e for e in st["rank0_assays"] if e.get("group_of", 1) > 1]) >= 6 * len(ckpts)     # several groups, padded members
    assert any(e.get("padded_T", e["T"]) > e["T"] for e in st["rank0_assays"])
    rb.main(rb.create_parser().parse_args(common + ["--dms-output", str(tmp_path / "single"), "--batch-short-tokens", "0"]))
    for r in rows:
        a = open(tmp_path / "grouped" / f"{r['DMS_id']}.csv").read()
        assert a == open(tmp_path / "single" / f"{r['DMS_id']}.csv").read(), r["DMS_id"]
        assert "nan" not in a.lower()


def test_cli_wt_marginals_overlapping_beyond_two_windows(lib, golden_dir, tmp_path):
    """wt-marginals, overlapping windows, on proteins of 1 023 ... 3 425 residues: two windows, the extra central window, the
    stepping loop (four / five / six windows) -- every branch of compute_fitness.py:433-475 -- against the reference CLI's
    scores (tests/golden/make_golden_wt_overlapping.py), through the single-assay CLI and the resident-model runner."""
    from proteingym_amd import run_benchmark as rb
    g = np.load(os.path.join(golden_dir, "golden_wt_overlapping.npz"))
    ck = os.path.join(golden_dir, "esm1v_toy_1.pt")
    rows = []
    for n_tok in g["n_tok"]:
        seq, muts = str(g[f"{n_tok}/seq"]), [str(m) for m in g[f"{n_tok}/mutants"]]
        pd.DataFrame({"mutant": muts, "DMS_score": np.zeros(len(muts))}).to_csv(tmp_path / f"W{n_tok}.csv", index=False)
        rows.append({"DMS_id": f"W{n_tok}", "DMS_filename": f"W{n_tok}.csv", "target_seq": seq, "file_length": len(seq)})
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    rb.main(rb.create_parser().parse_args(["--model-location", ck, "--model_type", "ESM1b", "--dms_mapping", str(tmp_path / "map.csv"),
                                           "--dms-input", str(tmp_path), "--dms-output", str(tmp_path / "runner"),
                                           "--scoring-strategy", "wt-marginals", "--scoring-window", "overlapping"]))
    for k, n_tok in enumerate(g["n_tok"]):
        want = g[f"{n_tok}/scores"]
        got = pd.read_csv(tmp_path / "runner" / f"W{n_tok}.csv", float_precision="round_trip")["esm1v_toy_1"].to_numpy()
        print(f"wt-marginals overlapping, {n_tok} tokens ({rb.wt_marginals_windows(int(n_tok), 'overlapping')} windows): max|err| {np.abs(got - want).max():.2e}")
        assert np.abs(got - want).max() < TOL
        if n_tok in (1538, 3000):                                    # the central-window branch also through the single-assay CLI
            _run_cli(["--model-location", ck, "--model_type", "ESM1b", "--dms-input", str(tmp_path), "--dms-output", str(tmp_path / "cli"),
                      "--scoring-strategy", "wt-marginals", "--scoring-window", "overlapping", "--dms_mapping", str(tmp_path / "map.csv"),
                      "--dms_index", str(k)])
            one = pd.read_csv(tmp_path / "cli" / f"W{n_tok}.csv", float_precision="round_trip")["esm1v_toy_1"].to_numpy()
            assert np.array_equal(one, got)
