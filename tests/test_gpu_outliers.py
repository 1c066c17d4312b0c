"""The default (f16x3) mode on a checkpoint with the features real ESM checkpoints are known for and the other synthetic
checkpoints lack (proteingym_amd/synthetic.py: outlier_weights): massive residual channels (+-3 000 from the first layer on and
again from a feed-forward bias), LayerNorm gains up to 30, heavy-tailed (Student-t, 3 d.o.f.) Linear weights -- at the ESM-1v
650M shape, full depth.  And the range guard: an activation beyond fp16's 65 504 must end in PGMI_EOVERFLOW (never in a wrong
number), after which the fp32 mode scores the same assay.

The CPU oracle (oracle/esm_oracle.py, pinned to the reference by tests/test_oracle_pinning.py) runs in fp32 like the reference
and once in fp64: with 3 000-sized residual channels (fp32 resolution 2.4e-4 there) the reference's own arithmetic is no longer
within 1e-4 of exact arithmetic on every row, so the bar is the north star's flat 1e-4 wherever the reference's fp32-vs-fp64
distance leaves room for it, and never more than that distance otherwise -- the test prints both."""
import numpy as np
import pytest

from proteingym_amd import _lib, esm as pesm, synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-4
L = 48


def _oracle(cfg, blob, seq, positions, dtype=None):
    import torch
    from oracle import esm_oracle as eo
    import frozen

    def compute():
        torch.set_num_threads(max(1, __import__("bench").usable_cores()))
        kw = {} if dtype is None else {"dtype": dtype}
        ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), **cfg, **kw)
        return {"table": eo.masked_marginals_table(ocfg, W, seq, positions=list(positions), batch=16)}
    fp = frozen.fingerprint([sorted(cfg.items()), seq, list(positions), str(dtype), blob]).hex()[:16]
    return frozen.cached(f"outliers_{cfg['layers']}_layers_{fp}", [sorted(cfg.items()), seq, list(positions), str(dtype), blob], compute)["table"]


@pytest.fixture(scope="module")
def outlier_case():
    import torch
    cfg = dict(synthetic.ESM1V_650M)
    blob = synthetic.outlier_weights(cfg, seed=5)
    seq, muts, _ = synthetic.random_assay(seed=31, L=L, n_single=300, n_multi=60)
    positions = sorted({int(one[1:-1]) for m in muts for one in m.split(":")})
    ref32 = _oracle(cfg, blob, seq, positions)
    ref64 = _oracle(cfg, blob, seq, positions, dtype=torch.float64)
    return cfg, blob, seq, muts, positions, ref32, ref64


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_outlier_checkpoint_650m_vs_oracle(lib, outlier_case, precision):
    from oracle import esm_oracle as eo
    cfg, blob, seq, muts, positions, ref32, ref64 = outlier_case
    model = pesm.EsmModel(cfg, blob, device=0, precision=precision)
    try:
        scores, table = pesm.Assay(model, seq, muts).run(want_table=True)
    except _lib.PgmiError as e:                       # the only acceptable alternative to a right answer
        assert precision == "f16x3" and "re-run with precision fp32" in str(e)
        pytest.skip("f16x3 reported PGMI_EOVERFLOW on the outlier checkpoint (the fp32 leg of this test is the product's answer)")
    finally:
        model.close()
    noise = float(np.abs(ref32[positions] - ref64[positions]).max())            # the reference's own fp32 arithmetic vs exact
    err32 = float(np.abs(table[positions] - ref32[positions]).max())
    err64 = float(np.abs(table[positions] - ref64[positions]).max())
    ref_s = np.array([eo.label_row(m, seq, ref32, 1) for m in muts])
    ref_s64 = np.array([eo.label_row(m, seq, ref64, 1) for m in muts])
    err_s, noise_s = float(np.abs(scores - ref_s).max()), float(np.abs(ref_s - ref_s64).max())
    rng_lp = float(ref32[positions].max() - ref32[positions].min())
    print(f"[{precision}] outlier checkpoint (650M shape, T={L + 2}): table rows max|err| {err32:.2e} vs the fp32 oracle, {err64:.2e} vs fp64; "
          f"scores {err_s:.2e}; the reference's own fp32-vs-fp64 distance: rows {noise:.2e}, scores {noise_s:.2e}; log-prob range {rng_lp:.1f}")
    assert np.isfinite(table[positions]).all() and np.isfinite(scores).all()
    assert err32 < max(TOL, 2.0 * noise) and err64 < max(TOL, 2.0 * noise)     # no further from exact arithmetic than the reference is
    assert err_s < max(TOL, 2.0 * noise_s)


def test_fp16_range_guard_raises_and_fp32_scores_the_assay(lib):
    """A feed-forward unit at 1e5 (its FC1 bias): the FC2 operand leaves fp16's range.  The f16x3 model must fail with
    PGMI_EOVERFLOW and the message that names the way out; the fp32 model then scores the assay."""
    from oracle import esm_oracle as eo
    cfg = dict(synthetic.ESM1V_650M, layers=6)
    blob = synthetic.random_weights(cfg, seed=9, embed_std=0.15)
    arrs = synthetic.blob_to_arrays(cfg, blob)
    arrs["layers.3.fc1.bias"][77] = 1.0e5
    seq, muts, _ = synthetic.random_assay(seed=33, L=40, n_single=120, n_multi=20)
    positions = sorted({int(one[1:-1]) for m in muts for one in m.split(":")})
    m16 = pesm.EsmModel(cfg, blob, device=0, precision="f16x3")
    with pytest.raises(_lib.PgmiError, match="re-run with precision fp32") as ei:
        pesm.Assay(m16, seq, muts).run()
    assert "non-finite" in str(ei.value)
    assert "libpgmi error -6" in str(ei.value)                    # PGMI_EOVERFLOW (include/pgmi.h)
    m32 = pesm.EsmModel(cfg, blob, device=0, precision="fp32")
    scores, table = pesm.Assay(m32, seq, muts).run(want_table=True)
    m32.close()
    ref = _oracle(cfg, blob, seq, positions)
    ref_s = np.array([eo.label_row(m, seq, ref, 1) for m in muts])
    err_t, err_s = float(np.abs(table[positions] - ref[positions]).max()), float(np.abs(scores - ref_s).max())
    print(f"overflow case, fp32 re-run: rows {err_t:.2e}, scores {err_s:.2e}")
    assert err_t < 5 * TOL and err_s < 5 * TOL            # a 1e5 activation: fp32's own resolution at that size is 8e-3 per add
    m16.close()


def test_run_benchmark_retries_an_overflowing_checkpoint_in_fp32_and_survives_a_bad_assay(lib, tmp_path):
    """The north-star runner on first contact with real data: two checkpoints, one of them with a
    feed-forward unit at 1e5 (PGMI_EOVERFLOW in f16x3), three assays, one of them with a wild-type mismatch in its file (the
    assertion of compute_fitness.py:243).  The runner must re-score the overflowing (assay, checkpoint) pairs on an fp32
    model it builds itself -- scores equal to an fp32 model's, bit for bit, and at fp32 parity against the oracle --, keep
    f16x3 for the clean checkpoint, write scores_summary.csv with precision_<checkpoint> = fp32, skip the bad assay's CSV and
    exit non-zero; both kinds of shard (whole assays incl. a short-assay group, position chunks)."""
    import pandas as pd
    from oracle import esm_oracle as eo
    from proteingym_amd import run_benchmark as rb
    cfg = dict(synthetic.ESM1V_650M, layers=4)
    clean = synthetic.random_weights(cfg, seed=9, embed_std=0.15)
    hot = clean.copy()
    synthetic.blob_to_arrays(cfg, hot)["layers.2.fc1.bias"][77] = 1.0e5
    synthetic.save_fair_esm_checkpoint(str(tmp_path / "clean_ck.pt"), cfg, clean)
    synthetic.save_fair_esm_checkpoint(str(tmp_path / "hot_ck.pt"), cfg, hot)
    rows, assays = [], {}
    for k, Ls in enumerate((40, 33, 52)):
        seq, muts, score = synthetic.random_assay(seed=40 + k, L=Ls, n_single=60, n_multi=15)
        if k == 2:
            muts[5] = ("A" if seq[2] != "A" else "C") + "3" + "W"
        pd.DataFrame({"mutant": muts, "DMS_score": score}).to_csv(tmp_path / f"R{k}.csv", index=False)
        rows.append({"DMS_id": f"R{k}", "DMS_filename": f"R{k}.csv", "target_seq": seq, "DMS_total_number_mutants": len(muts)})
        assays[f"R{k}"] = (seq, muts)
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    common = ["--model-location", str(tmp_path / "clean_ck.pt"), str(tmp_path / "hot_ck.pt"), "--model_type", "ESM1v",
              "--dms_mapping", str(tmp_path / "map.csv"), "--dms-input", str(tmp_path)]
    m16, _ = pesm.load_model_and_alphabet(str(tmp_path / "clean_ck.pt"), precision="f16x3")      # (the loader zeroes the <mask> row)
    m32, _ = pesm.load_model_and_alphabet(str(tmp_path / "hot_ck.pt"), precision="fp32")
    for mode, extra in (("assay", []), ("positions", ["--shard", "positions", "--chunk-forwards", "16"])):
        out = tmp_path / ("out_" + mode)
        with pytest.raises(SystemExit, match="1 assay.s. failed"):
            rb.main(rb.create_parser().parse_args(common + ["--dms-output", str(out)] + extra))
        assert not (out / "R2.csv").exists()
        summary = pd.read_csv(out / "scores_summary.csv", keep_default_na=False).set_index("DMS_id")
        assert list(summary["status"]) == ["ok", "ok", "failed"] and "does not match" in summary.loc["R2", "error"]
        for name in ("R0", "R1"):
            seq, muts = assays[name]
            got = pd.read_csv(out / f"{name}.csv", float_precision="round_trip")
            assert summary.loc[name, "precision_hot_ck"] == "fp32" and summary.loc[name, "precision_clean_ck"] == ""
            assert np.array_equal(got["clean_ck"].to_numpy(), pesm.Assay(m16, seq, muts).run())
            want32 = pesm.Assay(m32, seq, muts).run()
            assert np.array_equal(got["hot_ck"].to_numpy(), want32)
            assert np.array_equal(got["Ensemble_ESM1v"].to_numpy(), (got["clean_ck"].to_numpy() + want32) / 2)
    seq, muts = assays["R0"]
    positions = sorted({int(one[1:-1]) for m in muts for one in m.split(":")})
    hot_loaded = hot.copy()
    synthetic.blob_to_arrays(cfg, hot_loaded)["embed_tokens.weight"][32] = 0.0                  # pretrained.py:97
    ref = _oracle(cfg, hot_loaded, seq, positions)
    ref_s = np.array([eo.label_row(m, seq, ref, 1) for m in muts])
    err = float(np.abs(pd.read_csv(tmp_path / "out_assay" / "R0.csv")["hot_ck"].to_numpy() - ref_s).max())
    print(f"run_benchmark fp32 retry on the overflowing checkpoint: scores {err:.2e} from the fp32 oracle")
    assert err < 5 * TOL                                   # the bar of the fp32 re-run above (a 1e5 activation)
    m16.close()
    m32.close()


def test_run_indels_redoes_an_overflowing_library_in_fp32(lib, tmp_path):
    """Config 5's runner on the same kind of checkpoint: two indel libraries, a clean and a hot ESM2 checkpoint; the hot one
    leaves the fp16 range in f16x3, so both libraries are re-scored in fp32 for THAT checkpoint -- bit-equal to an fp32
    model's pseudo-ppl --, the clean one stays f16x3, scores_summary.csv says which; a library without the sequence column is
    dropped (no CSV, rc != 0) and the others are written."""
    import pandas as pd
    from proteingym_amd import run_indels as ri
    cfg = dict(synthetic.ESM2_650M, layers=3)
    clean = synthetic.random_weights(cfg, seed=19, embed_std=0.15)
    hot = clean.copy()
    synthetic.blob_to_arrays(cfg, hot)["layers.1.fc1.bias"][301] = 1.0e5
    synthetic.save_fair_esm_checkpoint(str(tmp_path / "esm2_clean_ck.pt"), cfg, clean)
    synthetic.save_fair_esm_checkpoint(str(tmp_path / "esm2_hot_ck.pt"), cfg, hot)
    libs = {}
    for k, (Lk, n) in enumerate(((37, 9), (52, 7))):
        _, seqs = synthetic.random_indel_library(seed=70 + k, L=Lk, n=n)
        libs[f"I{k}"] = list(seqs)
        pd.DataFrame({"mutant": [f"m{j}" for j in range(len(seqs))], "mutated_sequence": seqs, "DMS_score": np.arange(len(seqs)) * 0.5}
                     ).to_csv(tmp_path / f"I{k}.csv", index=False)
    pd.DataFrame({"mutant": ["m0"], "DMS_score": [0.0]}).to_csv(tmp_path / "I2.csv", index=False)           # no mutated_sequence column
    pd.DataFrame({"DMS_id": ["I0", "I1", "I2"], "DMS_filename": ["I0.csv", "I1.csv", "I2.csv"], "target_seq": ["M"] * 3}
                 ).to_csv(tmp_path / "map.csv", index=False)
    out = tmp_path / "out"
    with pytest.raises(SystemExit, match="1 assay.s. failed"):
        ri.main(ri.create_parser().parse_args(["--model-location", str(tmp_path / "esm2_clean_ck.pt"), str(tmp_path / "esm2_hot_ck.pt"), "--model_type", "ESM2",
                                               "--dms_mapping", str(tmp_path / "map.csv"), "--dms-input", str(tmp_path), "--dms-output", str(out)]))
    assert not (out / "I2.csv").exists()
    summary = pd.read_csv(out / "scores_summary.csv", keep_default_na=False).set_index("DMS_id")
    assert list(summary["status"]) == ["ok", "ok", "failed"] and "mutated_sequence" in summary.loc["I2", "error"]
    m16, alphabet = pesm.load_model_and_alphabet(str(tmp_path / "esm2_clean_ck.pt"), precision="f16x3")
    m32, _ = pesm.load_model_and_alphabet(str(tmp_path / "esm2_hot_ck.pt"), precision="fp32")
    mhot16, _ = pesm.load_model_and_alphabet(str(tmp_path / "esm2_hot_ck.pt"), precision="f16x3")
    for name, seqs in libs.items():
        got = pd.read_csv(out / f"{name}.csv", float_precision="round_trip")
        assert summary.loc[name, "precision_esm2_hot_ck"] == "fp32" and summary.loc[name, "precision_esm2_clean_ck"] == ""
        for col, model in (("esm2_clean_ck", m16), ("esm2_hot_ck", m32)):
            sl = pesm.SequenceLibrary(model, seqs, alphabet)
            want = sl.score()
            sl.close()
            assert np.array_equal(got[col].to_numpy(), want), (name, col)
        assert list(got.columns) == ["mutant", "mutated_sequence", "DMS_score", "esm2_clean_ck", "esm2_hot_ck"]        # ESM2: no ensemble column
    sl = pesm.SequenceLibrary(mhot16, libs["I0"], alphabet)
    with pytest.raises(pesm.PgmiError) as info:                                                             # what the runner caught
        sl.score()
    assert info.value.code == _lib.EOVERFLOW
    sl.close()
    # a call can hold hours of forwards: the range flag is read every 64 chunks, not only at the end of the call
    msmall, _ = pesm.load_model_and_alphabet(str(tmp_path / "esm2_hot_ck.pt"), precision="f16x3", max_rows=2048)     # 32 rows of <= 64 tokens per chunk
    sl = pesm.SequenceLibrary(msmall, synthetic.random_indel_library(seed=90, L=40, n=64)[1], alphabet)             # ~2 400 rows: 76 chunks
    with pytest.raises(pesm.PgmiError) as info:
        sl.score()
    assert info.value.code == _lib.EOVERFLOW and sl.stats()["batches"] == 64, sl.stats()
    sl.close()
    for m in (m16, m32, mhot16, msmall):
        m.close()
