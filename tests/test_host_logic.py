"""Host-side logic and the C-ABI surface -- no GPU needed (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import esm_oracle as eo, ref_harness as rh
from proteingym_amd import _lib, esm as pesm, synthetic, dist as pdist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_abi_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "pgmi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pgmi_[a-z0-9_]+)\s*\(", hdr))
    bound = {n for n, _, _ in _lib.SIGNATURES}
    assert declared == bound, (declared ^ bound)
    for n in declared:
        assert hasattr(lib, n)
    assert lib.pgmi_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback(lib):
    """Without a GPU the product path must fail loudly, not fall back."""
    if lib.pgmi_device_count() > 0:
        pytest.skip("GPU present")
    cfg = dict(synthetic.ESM1V_650M, layers=1, embed_dim=128, heads=2, ffn_dim=256)
    with pytest.raises(pesm.PgmiError, match="no HIP device|no CPU fallback"):
        pesm.EsmModel(cfg, synthetic.random_weights(cfg, 0))


def test_weight_count_and_config_validation(lib):
    cfg = dict(synthetic.ESM1V_650M)
    c = _lib.Config(abi_version=_lib.ABI_VERSION, vocab=33, precision=0, max_rows=0, **cfg)
    n = lib.pgmi_weight_count(C.byref(c))
    assert n == sum(int(np.prod(s)) for _, s in synthetic.key_shapes(cfg)) == 652355873 - 33 * 1280 + 33 * 1280
    c2 = dict(cfg, heads=16)                                   # head_dim 80: unsupported
    blob = np.zeros(8, np.float32)
    with pytest.raises(pesm.PgmiError):
        pesm.EsmModel(c2, blob)


def test_alphabet_matches_oracle_and_reference_vocab():
    a = pesm.Alphabet()
    assert a.all_toks == eo.ALL_TOKS and len(a) == 33
    assert (a.cls_idx, a.padding_idx, a.eos_idx, a.unk_idx, a.mask_idx) == (0, 1, 2, 3, 32)
    seq = "MKTAYIAKQXBZUO"
    _, _, t = a.get_batch_converter()([("p", seq), ("q", seq[:5])])
    assert np.array_equal(t[0], eo.tokenize(seq))
    assert t[1, 6] == a.eos_idx and (t[1, 7:] == a.padding_idx).all()
    assert a.get_idx("J") == a.unk_idx                        # get_idx (mutant letters, data.py:125-126) maps unknown -> <unk> ...
    with pytest.raises(KeyError):                             # ... the tokenizer (data.py:253-254) does not: 'J' is not in the vocabulary
        a.get_batch_converter()([("p", "MKJ")])
    assert a.encode("M K<mask>T") == [a.get_idx("M"), a.get_idx("K"), a.mask_idx, a.get_idx("T")]


def test_parse_mutants_matches_label_row_parsing(lib):
    seq, muts, _ = synthetic.random_assay(seed=3, L=50, n_single=40, n_multi=30, offset=5)
    sub_pos, sub_wt, sub_mt, off = pesm.parse_mutants(muts, seq, 5)
    k = 0
    for i, m in enumerate(muts):
        assert off[i] == k
        for s in m.split(":"):
            idx = int(s[1:-1]) - 5
            assert sub_pos[k] == 1 + idx and sub_wt[k] == eo.get_idx(s[0]) and sub_mt[k] == eo.get_idx(s[-1])
            k += 1
    assert off[-1] == k
    with pytest.raises(AssertionError, match="does not match"):
        pesm.parse_mutants([("C" if seq[0] != "C" else "A") + "5G"], seq, 5)
    for bad in ("A", "AG", "A5", "AxG", "A5G:", "A999G"):
        with pytest.raises((ValueError, AssertionError)):
            pesm.parse_mutants([bad], seq, 5)


def test_optimal_window_matches_oracle(lib):
    for n in (10, 1024, 1025, 1102, 2000, 3425):
        for i in list(range(0, n, 37)) + [n - 1, max(0, n - 513), 511, 512]:
            if 0 <= i < n:
                assert pesm.get_optimal_window(i, n, 1024) == eo.get_optimal_window(i, n, 1024)


def test_checkpoint_packing_and_key_checks(golden_dir, tmp_path):
    import torch
    cfg, sd = pesm._upgrade_state_dict(os.path.join(golden_dir, "esm1b_toy_lnb.pt"))
    assert cfg["emb_layer_norm_before"] == 1 and cfg["arch"] == _lib.ARCH_ESM1B
    blob = pesm.pack_state_dict(cfg, sd)
    assert blob.size == sum(int(np.prod(s)) for _, s in synthetic.key_shapes(cfg))
    bad = dict(sd)
    bad.pop("layers.0.fc1.bias")
    with pytest.raises(RuntimeError, match="Missing key"):
        pesm.pack_state_dict(cfg, bad)
    bad = dict(sd, extra=torch.zeros(1))
    with pytest.raises(RuntimeError, match="Unexpected key"):
        pesm.pack_state_dict(cfg, bad)
    cfg2, sd2 = pesm._upgrade_state_dict(os.path.join(golden_dir, "esm2_toy.pt"))
    assert cfg2["arch"] == _lib.ARCH_ESM2 and cfg2["ffn_dim"] == 4 * cfg2["embed_dim"]
    # synthetic checkpoints round-trip through the fair-esm file layout and the oracle loader
    c = dict(synthetic.ESM1V_650M, layers=1, embed_dim=128, heads=2, ffn_dim=256)
    b = synthetic.random_weights(c, 5)
    p = synthetic.save_fair_esm_checkpoint(str(tmp_path / "esm1v_syn.pt"), c, b)
    c3, sd3 = pesm._upgrade_state_dict(p)
    b3 = pesm.pack_state_dict(c3, sd3)
    exp = b.copy()
    exp[32 * 128:33 * 128] = 0                               # <mask> row zeroed at load (shared storage)
    assert np.array_equal(b3, exp)
    ocfg, W = eo.load_checkpoint(p)
    assert float(W["lm_head.weight"][32].abs().max()) == 0.0


def test_cli_parser_has_the_reference_flags():
    from proteingym_amd import compute_fitness as cf
    ours = {a.option_strings[0]: (a.nargs, a.default) for a in cf.create_parser()._actions if a.option_strings}
    frozen = ["--model_type", "--model-location", "--sequence", "--dms-input", "--dms_index", "--dms_mapping",
              "--mutation-col", "--dms-output", "--offset-idx", "--scoring-strategy", "--msa-path",
              "--msa-sampling-strategy", "--msa-samples", "--msa-weights-folder", "--seeds", "--filter-msa",
              "--hhfilter-min-cov", "--hhfilter-max-seq-id", "--hhfilter-min-seq-id", "--path-to-hhfilter",
              "--scoring-window", "--overwrite-prior-scores", "--target_seq", "--weight_file_name",
              "--MSA_start", "--MSA_end", "--nogpu"]
    for f in frozen:
        assert f in ours, f
    if rh.reference_available():
        ref = {a.option_strings[0]: (a.nargs, a.default)
               for a in rh.load_reference().create_parser()._actions if a.option_strings}
        for k, v in ref.items():
            assert ours[k] == v, (k, ours[k], v)


def test_lpt_partition_and_costs():
    shapes = synthetic.dms_shapes()
    assert len(shapes) == 217 and sum(s["seq_len"] + 2 for s in shapes) == 86613
    assert sum(s["n_total"] for s in shapes) == 2465767
    costs = [pdist.assay_cost(s["seq_len"]) for s in shapes]
    assert abs(sum(costs) / 83.0e15 - 1) < 0.02                # 83.0 PFLOP per checkpoint
    parts = pdist.lpt_partition(costs, 8)
    assert sorted(i for p in parts for i in p) == list(range(217))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) / (sum(loads) / 8) < 1.05
    assert abs(pdist.forward_flops(288) / 0.389e12 - 1) < 0.01  # 0.389 TFLOP per forward


def test_tranception_cli_input_resolution(tmp_path):
    """score_tranception_proteingym mirror: reference-file mode and manual mode resolve to the same inputs the
    reference computes (score_tranception_proteingym.py:49-85: MSA_start made 0-indexed, weight path optional)."""
    import pandas as pd
    from proteingym_amd import score_tranception_proteingym as st
    ref = tmp_path / "ref.csv"
    pd.DataFrame([{"DMS_id": "A1", "DMS_filename": "A1.csv", "target_seq": "mkvla", "MSA_filename": "A1.a2m",
                   "MSA_start": 2, "MSA_end": 5, "weight_file_name": "A1.npy"},
                  {"DMS_id": "B2", "DMS_filename": "B2.csv", "target_seq": "ACDE", "MSA_filename": "B2.a2m",
                   "MSA_start": 1, "MSA_end": 4, "weight_file_name": "B2.npy"}]).to_csv(ref, index=False)
    p = st.create_parser()
    a = p.parse_args(["--DMS_reference_file_path", str(ref), "--DMS_index", "0", "--inference_time_retrieval",
                      "--MSA_folder", "msas", "--MSA_weights_folder", "w"])
    assert st.resolve_inputs(a) == ("A1", "MKVLA", "A1.csv", ("msas/A1.a2m", "w/A1.npy", 1, 5))
    a = p.parse_args(["--DMS_reference_file_path", str(ref), "--DMS_index", "1", "--inference_time_retrieval", "--MSA_folder", "m"])
    assert st.resolve_inputs(a) == ("B2", "ACDE", "B2.csv", ("m/B2.a2m", None, 0, 4))
    a = p.parse_args(["--DMS_reference_file_path", str(ref), "--DMS_index", "1"])
    assert st.resolve_inputs(a) == ("B2", "ACDE", "B2.csv", None)
    a = p.parse_args(["--target_seq", "MKV", "--DMS_file_name", "X9.csv", "--inference_time_retrieval", "--MSA_folder", "m",
                      "--MSA_filename", "X9.a2m", "--MSA_start", "1", "--MSA_end", "3"])
    assert st.resolve_inputs(a) == ("X9", "MKV", "X9.csv", ("m/X9.a2m", None, 0, 3))
    with pytest.raises(NotImplementedError):
        st.main(p.parse_args(["--model_framework", "JAX"]))


def test_esm_cli_assay_resolution(tmp_path):
    """compute_fitness mirror, input resolution (compute_fitness.py:286-340): reference-file mode incl. the MSA
    Transformer's cropping of the target sequence to the alignment span, and manual mode."""
    import pandas as pd
    from proteingym_amd import compute_fitness as cf
    seq = "MKVLAAGIVGLTACDEFGHIK"
    pd.DataFrame({"mutant": ["K2A", "V3L:L4M"], "DMS_score": [0.1, -0.2]}).to_csv(tmp_path / "A1.csv", index=False)
    pd.DataFrame([{"DMS_id": "A1", "DMS_filename": "A1.csv", "target_seq": seq.lower(), "MSA_filename": "A1.a2m",
                   "MSA_start": 3, "MSA_end": 12, "weight_file_name": "A1.npy"}]).to_csv(tmp_path / "map.csv", index=False)
    p = cf.create_parser()
    common = ["--model-location", "x.pt", "--dms_index", "0", "--dms_mapping", str(tmp_path / "map.csv"),
              "--dms-input", str(tmp_path), "--dms-output", str(tmp_path / "out")]
    a = p.parse_args(common + ["--model_type", "ESM1v"])
    info = cf.resolve_assay(a)
    assert a.sequence == seq and info["first_position"] == 1 and info["mutant_col"] == "mutant"
    assert str(a.dms_output).endswith("out/A1.csv") and len(info["frame"]) == 2 and info["weight_file"] is None
    a = p.parse_args(common + ["--model_type", "MSA_transformer", "--msa-path", "msas", "--msa-weights-folder", "w"])
    info = cf.resolve_assay(a)
    assert a.sequence == seq[2:12] and info["msa_start"] == 3 and info["first_position"] == 3
    assert str(a.msa_path) == "msas/A1.a2m" and info["weight_file"] == "w/A1.npy"
    a = p.parse_args(["--model-location", "x.pt", "--model_type", "MSA_transformer", "--dms-input", str(tmp_path / "A1.csv"),
                      "--dms-output", str(tmp_path / "o2"), "--target_seq", seq, "--msa-path", "q.a2m"])
    info = cf.resolve_assay(a)
    assert a.sequence == seq and info["msa_start"] == 1 and (a.MSA_start, a.MSA_end) == (1, len(seq)) and info["dms_id"] == "A1"


def test_esm_cli_rejects_empty_assay_and_nogpu(tmp_path):
    """Error behaviour of the CLI mirror before any device work: an assay without rows raises the reference's
    ValueError (compute_fitness.py:343-344); --nogpu is refused loudly (there is no CPU path to fall back to)."""
    import pandas as pd
    from proteingym_amd import compute_fitness as cf
    pd.DataFrame({"mutant": [], "DMS_score": []}).to_csv(tmp_path / "E.csv", index=False)
    pd.DataFrame({"mutant": ["M1A"], "DMS_score": [0.0]}).to_csv(tmp_path / "F.csv", index=False)
    p = cf.create_parser()
    base = ["--model-location", "x.pt", "--model_type", "ESM1v", "--target_seq", "MKV", "--dms-output", str(tmp_path / "o")]
    with pytest.raises(ValueError, match="No rows found"):
        cf.main(p.parse_args(base + ["--dms-input", str(tmp_path / "E.csv")]))
    with pytest.raises(RuntimeError, match="GPU-only"):
        cf.main(p.parse_args(base + ["--dms-input", str(tmp_path / "F.csv"), "--nogpu"]))


def test_bench_cpu_baseline_leg_and_parity_field():
    """bench.py's CPU leg (the oracle timed on the host cores, bounded) on a tiny configuration: the returned object
    has the contract's fields, and the live parity check compares exactly the rows the baseline computed."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pgmi_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    sys.modules["pgmi_bench"] = bench
    spec.loader.exec_module(bench)
    assert 1 <= bench.usable_cores() <= (os.cpu_count() or 1)
    cfg = dict(synthetic.ESM1V_650M, layers=2, embed_dim=128, heads=2, ffn_dim=256)
    blob = synthetic.random_weights(cfg, seed=3)
    seq, muts, _ = synthetic.random_assay(seed=1, L=24, n_single=30, n_multi=0)
    ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), **cfg)
    table = eo.masked_marginals_table(ocfg, W, seq, positions=[4, 5, 6], batch=3)
    fake_gpu = np.full((len(seq) + 2, 33), np.nan, dtype=np.float32)
    fake_gpu[[4, 5, 6]] = table[[4, 5, 6]]
    fake_gpu[5, 7] += 3e-5
    base, parity = bench.cpu_baseline(cfg, blob, seq, len(muts), budget_s=2.0, gpu_table=fake_gpu)
    assert base["kind"] == "port" and base["unit"] == "mutants/s" and base["value"] > 0 and base["cores"] >= 1
    assert "2 of 2 layers" in base["sample"]
    assert parity["rows_compared"] == 3 and abs(parity["max_abs_err_vs_oracle"] - 3e-5) < 5e-6 and parity["tolerance"] == 1e-4


def test_checkpoint_loading_is_restricted_and_accepts_fused_in_proj(tmp_path, golden_dir):
    """(1) .pt files are unpickled with weights_only=True (+ argparse.Namespace): a file naming anything else is refused
    with a message that names the opt-in; (2) a legacy checkpoint with fused in_proj_weight / in_proj_bias loads to the
    same weight blob as its split twin (esm/multihead_attention.py:481-508)."""
    import torch
    from proteingym_amd import esm as pesm
    cfg, sd = pesm._upgrade_state_dict(os.path.join(golden_dir, "esm1v_toy_1.pt"))
    blob = pesm.pack_state_dict(cfg, sd)
    raw = pesm.load_checkpoint_file(os.path.join(golden_dir, "esm1v_toy_1.pt"))
    fused = {}
    for k, v in raw["model"].items():
        if ".self_attn.q_proj." in k:
            leaf = k.split(".")[-1]
            stem = k[: k.index("q_proj.")]
            fused[stem + "in_proj_" + leaf] = torch.cat([raw["model"][stem + n + "_proj." + leaf] for n in ("q", "k", "v")], 0)
        elif ".self_attn.k_proj." in k or ".self_attn.v_proj." in k:
            continue
        else:
            fused[k] = v
    assert any(k.endswith("in_proj_weight") for k in fused) and not any(".q_proj." in k for k in fused)
    path = str(tmp_path / "esm1v_legacy.pt")
    torch.save({"args": raw["args"], "model": fused}, path)
    cfg2, sd2 = pesm._upgrade_state_dict(path)
    assert cfg2 == cfg and np.array_equal(pesm.pack_state_dict(cfg2, sd2), blob)

    class Evil:
        def __reduce__(self):
            return (print, ("arbitrary code ran",))
    bad = str(tmp_path / "evil.pt")
    torch.save({"args": raw["args"], "model": {"x": Evil()}}, bad)
    with pytest.raises(RuntimeError, match="PGMI_UNSAFE_TORCH_LOAD"):
        pesm.load_checkpoint_file(bad)


# ---- run_benchmark: short assays scored several at a time (score_group) -------------------------------------------------
def test_short_assay_groups_plan():
    from proteingym_amd import run_benchmark as rb
    n_tok = {0: 42, 1: 60, 2: 44, 3: 199, 4: 90, 5: 43, 6: 150}
    groups = rb.plan_short_groups(list(n_tok), n_tok, group_rows=98304)
    flat = [i for g in groups for i in g]
    assert sorted(flat) == sorted(set(flat)) and set(flat) <= set(n_tok)
    for g in groups:
        assert len(g) > 1                                                     # a group of one is an ordinary assay
        assert [n_tok[i] for i in g] == sorted(n_tok[i] for i in g)           # sorted by length: padding to the last member
        assert sum(n_tok[i] for i in g) * max(n_tok[i] for i in g) <= 98304
    assert groups[0][:3] == [0, 5, 2]
    # a tight cap: every group still fits, nothing is lost except singletons
    tight = rb.plan_short_groups(list(n_tok), n_tok, group_rows=6000)
    assert all(sum(n_tok[i] for i in g) * max(n_tok[i] for i in g) <= 6000 for g in tight)
    assert rb.plan_short_groups([3], n_tok, 98304) == [] and rb.plan_short_groups([], n_tok, 98304) == []


class _GroupingFake:
    """A scorer whose score_group must give what score gives, member by member; records how it was called."""
    calls = []

    def __init__(self, location):
        self.salt = sum(map(ord, location))
        self.log = []

    def score(self, seq, mutants, offset):
        _GroupingFake.calls.append(("one", len(seq)))
        self.log.append(dict(seq_len=len(seq), rows=len(mutants), positions_run=1, T=len(seq) + 2, create_s=0.0, run_s=0.0))
        return np.array([((self.salt * 31 + len(seq) * 7 + sum(map(ord, m))) % 1000) / 37.0 for m in mutants])

    def score_group(self, assays):
        _GroupingFake.calls.append(("group", tuple(len(s) for s, _, _ in assays)))
        out = []
        for seq, mutants, offset in assays:
            out.append(np.array([((self.salt * 31 + len(seq) * 7 + sum(map(ord, m))) % 1000) / 37.0 for m in mutants]))
            self.log.append(dict(seq_len=len(seq), rows=len(mutants), positions_run=1, T=len(seq) + 2, create_s=0.0, run_s=0.0))
        return out

    def close(self):
        pass


def test_runner_groups_short_assays_and_keeps_every_csv(tmp_path):
    """Five assays, three of them short with few rows: the runner scores the long / many-row ones one at a time in its usual
    order, then the short ones as ONE group; every CSV holds its own rows' scores; the per-assay log names the right assay."""
    import pandas as pd
    from proteingym_amd import run_benchmark as rb
    rows, assays = [], {}
    for k, (L, n) in enumerate(((40, 30), (300, 50), (60, 25), (45, 400), (52, 20))):
        seq, muts, score = synthetic.random_assay(seed=20 + k, L=L, n_single=n, n_multi=4)
        pd.DataFrame({"mutant": muts, "DMS_score": score}).to_csv(tmp_path / f"S{k}.csv", index=False)
        rows.append({"DMS_id": f"S{k}", "DMS_filename": f"S{k}.csv", "target_seq": seq, "DMS_total_number_mutants": len(muts)})
        assays[f"S{k}"] = (seq, muts)
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    common = ["--model-location", "ckA.pt", "ckB.pt", "--model_type", "ESM1v", "--dms_mapping", str(tmp_path / "map.csv"), "--dms-input", str(tmp_path)]
    _GroupingFake.calls = []
    st = rb.main(rb.create_parser().parse_args(common + ["--dms-output", str(tmp_path / "g"), "--batch-short-rows", "100"]), make_model=_GroupingFake)
    per_ck = _GroupingFake.calls[: len(_GroupingFake.calls) // 2]
    assert per_ck == [("one", 45), ("one", 300), ("group", (40, 52, 60))]          # 404-row S3 is short but has too many rows for a group
    assert _GroupingFake.calls[len(per_ck):] == per_ck                          # same for the second checkpoint
    _GroupingFake.calls = []
    rb.main(rb.create_parser().parse_args(common + ["--dms-output", str(tmp_path / "n"), "--batch-short-tokens", "0"]), make_model=_GroupingFake)
    assert all(kind == "one" for kind, _ in _GroupingFake.calls)
    for name, (seq, muts) in assays.items():
        assert open(tmp_path / "g" / f"{name}.csv").read() == open(tmp_path / "n" / f"{name}.csv").read()
        got = pd.read_csv(tmp_path / "g" / f"{name}.csv", float_precision="round_trip")
        assert np.array_equal(got["ckA"].to_numpy(), _GroupingFake("ckA.pt").score(seq, muts, 1)) and list(got["mutant"]) == muts
    by_id = {(e["DMS_id"], e["checkpoint"]): e for e in st["rank0_assays"]}
    assert len(by_id) == 10 and all(by_id[(f"S{k}", c)]["seq_len"] == len(assays[f"S{k}"][0]) for k in range(5) for c in range(2))


LAUNCHERS = ["scoring_ESM1v_substitutions.sh", "scoring_ESM1b_substitutions.sh", "scoring_ESM2_substitutions.sh",
             "scoring_MSA_transformer_substitutions.sh", "scoring_Tranception_substitutions.sh",
             "scoring_Tranception_substitutions_no_retrieval.sh", "scoring_Tranception_indels_no_retrieval.sh", "scoring_Tranception_indels.sh"]


@pytest.mark.parametrize("script", LAUNCHERS)
def test_drop_in_launchers_read_the_zero_shot_config_and_build_a_valid_command_line(script, tmp_path):
    """scripts/scoring_DMS_zero_shot/*.sh: the launchers of the reference's names source a zero_shot_config.sh (here one written in
    the reference's variable names), and what they pass parses with this repository's CLI parsers -- and, where /root/reference
    exists, with the reference's own compute_fitness parser (same flags: a maintainer can swap the python line only)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg_dir = tmp_path / "scripts"
    (cfg_dir / "scoring_DMS_zero_shot").mkdir(parents=True)
    (tmp_path / "reference_files").mkdir()
    (cfg_dir / "zero_shot_config.sh").write_text(
        'export PROTEINGYM_CACHE="/data/pg"\n'
        'export DMS_data_folder_subs="${PROTEINGYM_CACHE}/DMS_ProteinGym_substitutions/"\n'
        'export DMS_data_folder_indels="${PROTEINGYM_CACHE}/DMS_ProteinGym_indels/"\n'
        'export DMS_MSA_data_folder="${PROTEINGYM_CACHE}/DMS_msa_files/"\n'
        'export DMS_MSA_weights_folder="${PROTEINGYM_CACHE}/DMS_msa_weights/"\n'
        'export DMS_reference_file_path_subs=../../reference_files/DMS_substitutions.csv\n'
        'export DMS_reference_file_path_indels=../../reference_files/DMS_indels.csv\n'
        'export DMS_output_score_folder_subs="${PROTEINGYM_CACHE}/zero_shot_substitutions_scores/"\n'
        'export DMS_output_score_folder_indels="${PROTEINGYM_CACHE}/zero_shot_indels_scores/"\n')
    env = dict(os.environ, ZERO_SHOT_CONFIG=str(cfg_dir / "zero_shot_config.sh"), PGMI_LAUNCH_ECHO="1", DMS_index="7")
    out = subprocess.run(["bash", os.path.join(root, "scripts", "scoring_DMS_zero_shot", script)], env=env, capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr
    argv = out.stdout.strip().split("\n")
    module, argv = argv[0], argv[1:]
    if module == "proteingym_amd.compute_fitness":
        from proteingym_amd import compute_fitness as cf
        a = cf.create_parser().parse_args(argv)
        assert a.dms_index == 7 and str(a.dms_input).rstrip("/") == "/data/pg/DMS_ProteinGym_substitutions"
        assert os.path.isabs(str(a.dms_mapping)) and str(a.dms_mapping).endswith("reference_files/DMS_substitutions.csv")
        assert str(a.dms_output).startswith("/data/pg/zero_shot_substitutions_scores/")
        if "ESM1v" in script:
            assert len(a.model_location) == 5 and "ESM1v" in a.model_type and a.scoring_strategy == "masked-marginals"
        if "ESM1b" in script:
            assert a.scoring_strategy == "wt-marginals" and a.scoring_window == "overlapping"
        if "MSA_transformer" in script:
            assert list(a.seeds) == [1, 2, 3, 4, 5] and str(a.msa_weights_folder).endswith("DMS_msa_weights_for_MSA_Transformer")
        if rh.reference_available():
            b = rh.load_reference().create_parser().parse_args(argv)
            assert vars(b) == {k: v for k, v in vars(a).items() if k in vars(b)}
    else:
        assert module == "proteingym_amd.score_tranception_proteingym"
        from proteingym_amd import score_tranception_proteingym as tcli
        a = tcli.create_parser().parse_args(argv)
        assert a.DMS_index == 7 and bool(a.indel_mode) == ("indels" in script) and bool(a.inference_time_retrieval) == ("no_retrieval" not in script)
        assert a.DMS_reference_file_path.endswith("DMS_indels.csv" if "indels" in script else "DMS_substitutions.csv")
        assert bool(a.clustal_omega_location) == (script == "scoring_Tranception_indels.sh")      # indels WITH retrieval: the aligner


@pytest.mark.parametrize("script", ["scoring_Tranception.sh", "scoring_Tranception_indels.sh", "scoring_ESM1b_substitutions.sh"])
def test_clinical_launchers_build_a_valid_command_line(script, tmp_path):
    """scripts/scoring_clinical_zero_shot/*.sh read the clinical_* variables of zero_shot_config.sh."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg_dir = tmp_path / "scripts"
    (cfg_dir / "scoring_clinical_zero_shot").mkdir(parents=True)
    (cfg_dir / "zero_shot_config.sh").write_text(
        'export clinical_reference_file_path_subs=/data/pg/reference_files/clinical_substitutions.csv\n'
        'export clinical_data_folder_subs=/data/pg/clinical_ProteinGym_substitutions\n'
        'export clinical_MSA_data_folder_subs=/data/pg/clinical_msa_files\n'
        'export clinical_MSA_weights_folder_subs=/data/pg/clinical_msa_weights\n'
        'export clinical_output_score_folder_subs=/data/pg/zero_shot_clinical_substitutions_scores\n'
        'export clinical_reference_file_path_indels=/data/pg/reference_files/clinical_indels.csv\n'
        'export clinical_data_folder_indels=/data/pg/clinical_ProteinGym_indels\n'
        'export clinical_MSA_data_folder_indels=/data/pg/clinical_msa_files_indels\n'
        'export clinical_MSA_weights_folder_indels=/data/pg/clinical_msa_weights_indels\n'
        'export clinical_output_score_folder_indels=/data/pg/zero_shot_clinical_indels_scores\n')
    env = dict(os.environ, ZERO_SHOT_CONFIG=str(cfg_dir / "zero_shot_config.sh"), PGMI_LAUNCH_ECHO="1", DMS_index="11")
    out = subprocess.run(["bash", os.path.join(root, "scripts", "scoring_clinical_zero_shot", script)], env=env, capture_output=True, text=True,
                         cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr
    module, *argv = out.stdout.strip().split("\n")
    if script.startswith("scoring_Tranception"):
        from proteingym_amd import score_tranception_proteingym as tcli
        assert module == "proteingym_amd.score_tranception_proteingym"
        a = tcli.create_parser().parse_args(argv)
        indels = "indels" in script
        assert a.DMS_index == 11 and a.inference_time_retrieval and bool(a.indel_mode) == indels and bool(a.clustal_omega_location) == indels
        assert a.DMS_reference_file_path.endswith("clinical_indels.csv" if indels else "clinical_substitutions.csv")
        assert a.MSA_folder == ("/data/pg/clinical_msa_files_indels" if indels else "/data/pg/clinical_msa_files")
        assert a.MSA_weights_folder == ("/data/pg/clinical_msa_weights_indels" if indels else "/data/pg/clinical_msa_weights")
        assert a.output_scores_folder.endswith("Tranception/Tranception_L")
    else:
        from proteingym_amd import run_benchmark as rb
        assert module == "proteingym_amd.run_benchmark"
        a = rb.create_parser().parse_args(argv)
        assert a.scoring_strategy == "wt-marginals" and a.scoring_window == "overlapping"
        assert str(a.dms_mapping).endswith("clinical_substitutions.csv")


def test_product_path_never_touches_the_oracle():
    """oracle/ is test infrastructure: no module of the product package names it, bench.py only inside its cpu_baseline leg,
    __graft_entry__ only in build() (compiling the checker) and smoke() (checking against it)."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        hits = []
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            if any(n == "oracle" or n.startswith("oracle.") for n in names):
                hits.append(node.lineno)
        return tree, hits
    pkg = os.path.join(root, "proteingym_amd")
    for fn in sorted(os.listdir(pkg)):
        if fn.endswith(".py"):
            assert oracle_imports(os.path.join(pkg, fn))[1] == [], fn
    for fn in sorted(os.listdir(os.path.join(pkg, "csrc"))):
        if fn.endswith((".hip", ".h")):
            assert "oracle" not in open(os.path.join(pkg, "csrc", fn)).read(), fn
    tree, hits = oracle_imports(os.path.join(root, "bench.py"))
    spans = {f.name: (f.lineno, f.end_lineno) for f in ast.walk(tree) if isinstance(f, ast.FunctionDef)}
    lo, hi = spans["cpu_baseline"]
    assert hits and all(lo <= h <= hi for h in hits), (hits, spans["cpu_baseline"])
    tree, hits = oracle_imports(os.path.join(root, "__graft_entry__.py"))
    spans = {f.name: (f.lineno, f.end_lineno) for f in ast.walk(tree) if isinstance(f, ast.FunctionDef)}
    assert all(any(lo <= h <= hi for lo, hi in (spans["build"], spans["smoke"])) for h in hits), hits


def test_bench_box_state_never_raises(monkeypatch):
    """bench.box_state reports the shader clock / socket power beside the line; without a GPU (or rocm-smi) it must say
    'unavailable' -- and a step that fails must not take the bench line down either."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = [0]

    def step():
        calls[0] += 1
        time.sleep(0.005)
    out = bench.box_state(step, seconds=0.3)
    assert calls[0] > 0 and ("unavailable" in out or {"sclk_mhz", "socket_power_w", "samples"} <= set(out))
    monkeypatch.setenv("PATH", "/nonexistent")                       # no rocm-smi at all
    assert "unavailable" in bench.box_state(step, seconds=0.1)


def test_bench_line_carries_the_metric_as_survey_8d_words_it():
    """bench.end_to_end_fields: the top-level keys beside the device-only `value` -- value_end_to_end (the BLAT-shaped assay from mutant
    strings to a CSV, weights resident), the whole table's end-to-end rate and the one-GPU point of the N > 1 line's strong pass;
    bench.ffn_traffic prefers counters collected in the run and says where its numbers come from; the N > 1 line is the same step."""
    import importlib.util
    root = os.path.join(os.path.dirname(__file__), "..")
    spec = importlib.util.spec_from_file_location("bench_mod_fields", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    e2e = {"mutants": 4996, "seconds": 0.32, "assay_create_s": 0.01, "run_s": 0.29, "csv_s": 0.02}
    b217 = {"mutants": 2465767, "seconds": 165.0, "rank0_wall_clock": {"assay_run_s": 160.0}}
    f = bench.end_to_end_fields(e2e, b217)
    assert abs(f["value_end_to_end"] - 4996 / 0.32) < 1e-9 and f["value_end_to_end_detail"] is e2e
    assert abs(f["benchmark_217_end_to_end_mutants_per_s"] - 2465767 / 165.0) < 1e-9
    assert abs(f["one_gpu_same_workload_mutants_per_s"] - 2465767 / 160.0) < 1e-9 and "strong_scaling_217" in f["one_gpu_same_workload"]
    assert bench.end_to_end_fields(None, None) == {} and "value_end_to_end" not in bench.end_to_end_fields({"error": "x"}, None)
    assert bench.live_traffic("fp32") == {"unavailable": "f16x3 only"}
    M, D, F = 82368, 1280, 5120
    live = {"lib_digest": "abc", "git_head": None, "rows_per_launch": None,
            "kernels": {"void pgmi::gemm16x_kernel<1, 1, false>": {"dispatches": 2, "fetch_bytes": 3.3e9, "write_bytes": 1.69e9},
                        "void pgmi::gemm16x_kernel<0, 0, false>": {"dispatches": 4, "fetch_bytes": 2.0e9, "write_bytes": 0.42e9}}}
    traffic, detail = bench.ffn_traffic("f16x3", M, D, F, live=live)
    assert traffic == 3.3e9 + 1.69e9 and "bench.py itself" in detail["source"] and detail["fc1"]["fetch_over_algorithmic"] > 7
    none, why = bench.ffn_traffic("f16x3", M, D, F, live={"unavailable": "rocprofv3 not on this box"})
    assert none is None or "committed" in why["source"]              # falls back to the committed file (or says why it cannot)
    src = open(os.path.join(root, "bench.py")).read()
    for key in ('"value_is"', '"strong_scaling_217"', '"scaling": "weak"', "value_end_to_end"):
        assert key in src
    assert '"scaling": "strong"' not in src                           # the N > 1 headline is the N = 1 step on every rank


def test_tranception_intermediate_roots_for_multi_mutants():
    """tranception.TranceptionModel.intermediate_roots: in a pairwise library the double mutants are served by "wild type + first substitution"
    roots (appended to the call unless that single mutant is a row already), every reference is a root, every sequence equals its root before
    its first difference from it, and fewer rows are forwarded; a library of singles, or groups too small to pay for a root, stay as they are."""
    from proteingym_amd import tranception as ptr
    rng = np.random.default_rng(0)
    T = 60
    wt = rng.integers(5, 25, size=T).astype(np.int32)
    wt[0], wt[-1] = 1, 2

    def sub(s, p, tok):
        s = s.copy()
        s[p] = tok if s[p] != tok else (tok - 5 + 1) % 20 + 5
        return s
    rows = [wt.copy()]
    for i in (3, 10, 20):
        for a in (5, 6):
            for j in (30, 40, 50):
                for b in (7, 8, 9):
                    rows.append(sub(sub(wt, i, a), j, b))
    rows.append(sub(wt, 3, 5))                                        # one of the intermediates is a row of the library
    ids = np.stack(rows).astype(np.int32)
    B = len(ids)
    ref = np.zeros(B, dtype=np.int32)
    ids_c, ref_c, extra = ptr.TranceptionModel.intermediate_roots(ids, ref)
    assert extra == 5 and ids_c.shape == (B + 5, T) and np.array_equal(ids_c[:B], ids)
    full = ref_c.copy()
    assert (full[B:] == 0).all()                                       # made from the wild-type row (the caller copies its retrieval arguments)
    full[B:] = np.arange(B, B + extra)
    assert (full[full] == full).all()
    assert full[B - 1] == B - 1                                        # the listed single mutant serves its doubles: forwarded in full now

    def first_diff(b):
        d = np.flatnonzero(ids_c[b] != ids_c[full[b]])
        return int(d[0]) if d.size else T - 1
    before = T + sum(T - int(np.flatnonzero(ids[b] != wt)[0]) for b in range(1, B))
    after = sum(T if full[b] == b else T - first_diff(b) for b in range(B + extra))
    assert after < 0.6 * before
    for b in range(1, B - 1):                                          # every double starts at its SECOND substitution
        assert first_diff(b) == int(np.flatnonzero(ids[b] != wt)[1])
    singles = np.stack([wt] + [sub(wt, p, 7) for p in range(1, 40)]).astype(np.int32)
    same = ptr.TranceptionModel.intermediate_roots(singles, np.zeros(len(singles), dtype=np.int32))
    assert same[2] == 0 and same[0] is singles
    few = np.stack([wt, sub(sub(wt, 50, 5), 55, 6), sub(sub(wt, 50, 5), 57, 6)]).astype(np.int32)     # two members save 5 + 7 rows: less than a root costs
    assert ptr.TranceptionModel.intermediate_roots(few, np.zeros(3, dtype=np.int32))[2] == 0


def test_pgmi_score_mutants_is_label_rows_arithmetic_on_the_host():
    """pgmi_score_mutants (host C, no GPU): per substitution an f32 difference table[pos, mt] - table[pos, wt], accumulated in
    double in the order of the mutation string (compute_fitness.py:245-250) -- checked against a plain python loop of that
    arithmetic, bit for bit, incl. depth-5 multi-mutants, an empty column and NaN (never computed) rows; indices outside the
    table are refused."""
    from proteingym_amd import esm as pesm, synthetic, _lib
    seq, muts, _ = synthetic.random_assay(seed=4, L=57, n_single=200, n_multi=120)
    rng = np.random.default_rng(0)
    table = (rng.standard_normal((len(seq) + 2, 33)) * 7).astype(np.float32)
    table[5] = np.nan
    sub_pos, sub_wt, sub_mt, mut_off = pesm.parse_mutants(muts, seq, 1)
    got = pesm.score_from_table(table, muts, seq, 1)
    want = np.zeros(len(muts))
    for i in range(len(muts)):
        acc = 0.0
        for k in range(mut_off[i], mut_off[i + 1]):
            acc += float(np.float32(table[sub_pos[k], sub_mt[k]] - table[sub_pos[k], sub_wt[k]]))
        want[i] = acc
    assert np.array_equal(got, want, equal_nan=True) and np.isnan(got).any() and np.isfinite(got).sum() > 200
    assert pesm.score_parsed(table, sub_pos[:0], sub_wt[:0], sub_mt[:0], np.zeros(1, np.int64)).shape == (0,)
    bad = sub_pos.copy()
    bad[3] = len(seq) + 2
    with pytest.raises(_lib.PgmiError, match="reads table"):
        pesm.score_parsed(table, bad, sub_wt, sub_mt, mut_off)


def test_workload_scale_projections_of_configs_4_and_5():
    """bench.py's secondary.tranception_217_projection / indels_projection (scripts/bench_projection.py): the planning arithmetic on
    the real tables -- the sample covers every protein-length bin, the projection conserves mutants and seconds, scales linearly in the
    measured unit costs, reproduces SURVEY 8f's totals (~2 700 PFLOP for config 4's reference loop, 1.95e8 masked forwards / ~2e5 PFLOP
    for config 5), and the N = 8 figures come from the product planners (max/mean close to 1)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import bench_projection as bp
    from proteingym_amd import synthetic
    shapes = synthetic.dms_shapes()
    sample = bp.tranception_sample(shapes)
    assert [e["bin"] for e in sample] == list(range(len(bp.LENGTH_BINS)))
    assert all(bp.bin_of(e["seq_len"]) == e["bin"] and 1 <= e["singles"] <= 384 for e in sample)
    assert sample[-1]["multis"] == 0 and all(e["multis"] == 384 for e in sample[:4])          # no multi-mutants beyond the 1 022-residue context
    unit = {(e["bin"], kind): 2e-9 * (1 + e["bin"]) * (0.5 if kind == "multi" else 1.0) for e in sample for kind in ("single", "multi")}
    p = bp.project_tranception(shapes, unit)
    assert p["mutants"] == 2465767 and sum(b["mutants"] for b in p["by_length_bin"].values()) == 2465767
    assert abs(sum(b["seconds"] for b in p["by_length_bin"].values()) - p["seconds_1_gpu"]) < 1e-9 * p["seconds_1_gpu"]
    assert p["seconds_1_gpu"] / 8 <= p["seconds_8_gpus_planned"] < 1.1 * p["seconds_1_gpu"] / 8
    assert 1.0 <= p["planned_load_max_over_mean_8_gpus"] < 1.1
    p2 = bp.project_tranception(shapes, {k: 3 * v for k, v in unit.items()})
    assert abs(p2["seconds_1_gpu"] - 3 * p["seconds_1_gpu"]) < 1e-9 * p2["seconds_1_gpu"]
    pflop = sum(bp.tranception_flops(s["seq_len"], s["n_total"]) for s in shapes) / 1e15
    assert 2500 < pflop < 3000                                                                  # ~2 700 PFLOP
    ish = synthetic.indel_shapes()
    fw, fl = bp.indel_forwards(ish)
    assert len(ish) == 66 and sum(s["n_total"] for s in ish) == 287207 and 1.9e8 < fw < 2.0e8 and 1.8e20 < fl < 2.2e20
    Ls = bp.indel_sample_lengths(ish)
    assert Ls[0] == min(s["seq_len"] for s in ish) and Ls[-1] == max(s["seq_len"] for s in ish) and 735 in Ls and len(Ls) >= 4
    q = bp.project_indels(ish, {L: 2.0e-6 * (L + 2) for L in Ls})
    assert q["masked_forwards"] == fw and q["largest_assay"]["DMS_id"].startswith("CAPSD_AAV2S") and q["largest_assay"]["share_of_seconds"] > 0.8
    assert q["seconds_1_gpu"] / 8 <= q["seconds_8_gpus_planned"] < 1.01 * q["seconds_1_gpu"] / 8
    exact = sum(max(0, s["seq_len"] - 2) * s["n_total"] * 2.0e-6 * (s["seq_len"] + 2) for s in ish)
    assert abs(q["seconds_1_gpu"] - exact) < 0.2 * exact                                         # interpolation in FLOPs per forward between the sampled lengths


def test_parser_agrees_with_label_row_on_arbitrary_strings(lib):
    """pgmi_parse_mutants against the parsing of label_row (compute_fitness.py:240-250: ``row.split(":")``, ``mutation[0]``,
    ``int(mutation[1:-1]) - offset_idx``, ``mutation[-1]``, the wild-type assertion) on strings drawn from the characters that
    matter: whenever the C parser accepts a string, python's own parsing accepts it too and gives the same (position, wt, mt)
    triples; everything python would mis-read silently (a negative index wraps around the sequence) or cannot index is refused."""
    from hypothesis import given, settings, strategies as st
    seq = "MKTAYIAKQRQISFVKSHFSRQ"
    offset = 3

    @settings(max_examples=600, deadline=None)
    @given(st.lists(st.text(alphabet="MKTAYIQRSFVHGC:0123456789-+ _x", min_size=0, max_size=12), min_size=1, max_size=4))
    def check(muts):
        try:
            sub_pos, sub_wt, sub_mt, off = pesm.parse_mutants(muts, seq, offset)
        except (ValueError, AssertionError):
            return
        k = 0
        for i, m in enumerate(muts):
            assert off[i] == k
            for s in m.split(":"):
                idx = int(s[1:-1]) - offset                              # python accepts what the C parser accepted ...
                assert 0 <= idx < len(seq) and seq[idx] == s[0]          # ... inside the sequence, at the listed wild type
                assert (sub_pos[k], sub_wt[k], sub_mt[k]) == (1 + idx, eo.get_idx(s[0]), eo.get_idx(s[-1]))
                k += 1
        assert off[-1] == k
    check()


def test_tranception_slices_agree_with_the_oracle_on_drawn_libraries():
    """tranception.get_sequence_slices (scoring_utils.py:152-203 on plain lists) against the oracle's pandas restatement on drawn inputs:
    protein lengths on both sides of a small context, barycentres of multi-mutants at the window edges, duplicated rows, the wild type
    among the rows, 'optimal' / 'sliding', indel libraries of mixed lengths -- the same frame, row for row, in the same order."""
    import pandas as pd
    from hypothesis import given, settings, strategies as st
    from oracle import tranception_oracle as to
    from proteingym_amd import tranception as ptr

    @settings(max_examples=150, deadline=None)
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
                muts.append(muts[0])                                      # a duplicated row
            df = pd.DataFrame({"mutant": muts, "mutated_sequence": [ptr.get_mutated_sequence(wt, m, start_idx) for m in muts]})
        kw = dict(start_idx=start_idx, scoring_window="sliding" if mode == "sliding" else "optimal", indel_mode=mode == "indel")
        a = ptr.get_sequence_slices(df.copy(), wt, ctx, **kw).reset_index(drop=True)
        b = to.get_sequence_slices(df.copy(), wt, ctx, **kw).reset_index(drop=True)
        cols = ["mutated_sequence", "sliced_mutated_sequence", "window_start", "window_end"]
        assert list(a.columns) == list(b.columns), (mode, list(a.columns), list(b.columns))
        assert len(a) == len(b) and all(list(a[c]) == list(b[c]) for c in cols), (seed, mode, L, ctx)
    check()
