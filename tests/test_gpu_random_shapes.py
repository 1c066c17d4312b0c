"""A seeded sweep over shapes nobody picked by hand: architecture (ESM-1b/1v with and without the LayerNorm before the stack,
with and without token dropout; ESM2), depth, heads x head dimension, feed-forward width, protein length (1 residue ... beyond
the 1 024-token window), residue numbering offset, mutant libraries (singles, multi-mutants that repeat positions), the three
strategies of compute_fitness.py (masked-marginals :486-504, wt-marginals :433-475 with both window modes, pseudo-ppl :258-279)
and both parity-gated precisions -- every case through the C ABI against the CPU oracle (oracle/esm_oracle.py, pinned to the
reference by tests/test_oracle_pinning.py) at the flat 1e-4.  Widths are narrow (the oracle runs live, a few seconds in all);
what the sweep varies is everything that does not depend on the width: tile edges (T % 32, T < 32, one tile, many), the window
logic, the ragged batches of the pseudo-ppl packer, the parser.

PGMI_SWEEP_CASES (default 36) and PGMI_SWEEP_SEED (default 2026) widen or move the sweep:
    PGMI_SWEEP_CASES=400 PGMI_SWEEP_SEED=7 python -m pytest tests/test_gpu_random_shapes.py -m gpu -q -s"""
import os

import numpy as np
import pytest

from proteingym_amd import _lib, compute_fitness as pcf, esm as pesm, synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-4
CASES = int(os.environ.get("PGMI_SWEEP_CASES", "36"))
SEED = int(os.environ.get("PGMI_SWEEP_SEED", "2026"))


def _draw(rng):
    arch = rng.choice(["esm1b", "esm1b_lnb", "esm1b_nodrop", "esm2"])
    heads = int(rng.choice([1, 2, 3, 4, 5]))
    head_dim = int(rng.choice([16, 32, 64, 64]))
    if arch == "esm2" and rng.random() < 0.2:
        head_dim = 24                                                    # ESM2-35M's
    precision = str(rng.choice(["f16x3", "f16x3", "fp32"]))
    if (heads * head_dim) % 32:                                          # pgmi_model_create: embed_dim and ffn_dim are multiples of 32 (every
        heads = heads * 2 if head_dim == 16 else 4                       # released width is: 320 / 480 / 640 / 768 / 1280 / 2560 / 5120)
    D = heads * head_dim
    ffn = int(rng.choice([2, 3, 4])) * D
    base = synthetic.ESM2_650M if arch == "esm2" else synthetic.ESM1V_650M
    cfg = dict(base, layers=int(rng.integers(1, 4)), embed_dim=D, heads=heads, ffn_dim=ffn)
    if arch == "esm1b_lnb":
        cfg["emb_layer_norm_before"] = 1
    if arch == "esm1b_nodrop":
        cfg["token_dropout"] = 0
    kind = rng.choice(["tiny", "short", "tile_edge", "medium", "window"], p=[0.12, 0.3, 0.25, 0.25, 0.08])
    L = {"tiny": lambda: int(rng.integers(1, 6)), "short": lambda: int(rng.integers(6, 62)),
         "tile_edge": lambda: int(rng.choice([30, 62, 94, 126, 222, 254])) + int(rng.integers(-1, 2)),
         "medium": lambda: int(rng.integers(62, 420)), "window": lambda: int(rng.integers(1023, 1100))}[kind]()
    return dict(arch=arch, cfg=cfg, precision=precision, L=L, offset=int(rng.choice([1, 1, 2, 17, 290])),
                strategy=str(rng.choice(["masked-marginals", "masked-marginals", "wt-marginals", "pseudo-ppl"])),
                window=str(rng.choice(["optimal", "overlapping"])), seed=int(rng.integers(1 << 30)))


def _library(rng, seq, offset, n):
    """Singles and multi-mutants (depth 2-5, positions may repeat inside a mutant: label_row adds every listed substitution)."""
    aa = synthetic.AA
    out = []
    for _ in range(n):
        depth = 1 if rng.random() < 0.6 else int(rng.integers(2, 6))
        subs = []
        for _ in range(depth):
            p = int(rng.integers(0, len(seq)))
            subs.append(f"{seq[p]}{p + offset}{rng.choice([a for a in aa if a != seq[p]])}")
        out.append(":".join(subs))
    return out


@pytest.mark.parametrize("case", range(CASES))
def test_random_shape_vs_oracle(lib, case):
    from oracle import esm_oracle as eo
    rng = np.random.default_rng([SEED, case])
    c = _draw(rng)
    cfg, L = c["cfg"], c["L"]
    if c["strategy"] == "pseudo-ppl":
        L = min(max(L, 3), 90)                                           # (L - 2) forwards per sequence on the CPU
    blob = synthetic.random_weights(cfg, seed=c["seed"] % 100000, embed_std=0.3)
    ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), **cfg)
    model = pesm.EsmModel(cfg, blob, device=0, precision=c["precision"])
    alphabet = pesm.Alphabet()
    seq = synthetic.random_sequence(rng, L)
    what = f"case {case}: {c['arch']} {cfg['layers']}x{cfg['embed_dim']} ({cfg['heads']} heads) ffn {cfg['ffn_dim']} {c['precision']} " \
           f"L={L} offset={c['offset']} {c['strategy']}"
    try:
        if c["strategy"] == "masked-marginals":
            muts = _library(rng, seq, c["offset"], int(rng.integers(1, 40)))
            assay = pesm.Assay(model, seq, muts, offset_idx=c["offset"], alphabet=alphabet)
            scores, table = assay.run(want_table=True)
            positions = [int(p) for p in assay.positions]
            assay.close()
            assert positions == sorted({int(s[1:-1]) - c["offset"] + 1 for m in muts for s in m.split(":")}), what
            ref = eo.masked_marginals_table(ocfg, W, seq, positions=positions, batch=8 if L + 2 <= 1024 else 1)
            err_t = float(np.abs(table[positions] - ref[positions]).max())
            ref_s = np.array([eo.label_row(m, seq, ref, c["offset"]) for m in muts])
        elif c["strategy"] == "wt-marginals":
            if cfg["arch"] == _lib.ARCH_ESM1B and L + 2 > 1024 and c["window"] == "optimal":
                with pytest.raises(pesm.PgmiError, match="maximum sequence length"):     # modules.py:256-260: the reference raises too
                    pcf.wt_marginals_table(model, alphabet, seq, "optimal")
                return
            muts = _library(rng, seq, c["offset"], int(rng.integers(1, 40)))
            table = pcf.wt_marginals_table(model, alphabet, seq, c["window"])[0]
            scores = pesm.score_from_table(table, muts, seq, c["offset"])
            ref = eo.wt_marginals_table(ocfg, W, seq, c["window"])
            err_t = float(np.abs(table - ref).max())
            ref_s = np.array([eo.label_row(m, seq, ref, c["offset"]) for m in muts])
            what += f" window={c['window']}"
        else:
            _, seqs = synthetic.random_indel_library(seed=c["seed"] % 9973, L=L, n=int(rng.integers(1, 6)), max_edit=min(3, max(1, L - 3)))
            seqs = [s for s in seqs if len(s) >= 1]
            sl = pesm.SequenceLibrary(model, seqs, alphabet)
            scores, terms = sl.score(want_terms=True)
            sl.close()
            ref_s = np.array([eo.compute_pppl(ocfg, W, s) for s in seqs])
            err_t = 0.0
            for s, t in zip(seqs, terms):
                assert len(t) == max(0, len(s) - 2), what
            scores = np.asarray(scores)
            # a sum of up to 90 terms: each term at the flat bar
            assert np.all(np.abs(scores - ref_s) <= TOL * np.maximum(1, [len(s) - 2 for s in seqs])), (what, scores, ref_s)
            print(what, f"{len(seqs)} sequences, sums max|err| {float(np.abs(scores - ref_s).max()):.2e}")
            return
        err_s = float(np.abs(np.asarray(scores) - ref_s).max())
        print(what, f"rows max|err| {err_t:.2e}, scores max|err| {err_s:.2e} ({len(muts)} mutants)")
        assert err_t < TOL, what
        assert err_s < TOL, what
    finally:
        model.close()


def _draw_tranception(rng):
    heads = int(rng.choice([4, 8, 12]))                                  # grouped ALiBi / convolutions: heads in four groups, 64 wide
    D = heads * 64
    cfg = dict(synthetic.TRANCEPTION_L, layers=int(rng.integers(1, 4)), embed_dim=D, heads=heads, ffn_dim=int(rng.choice([2, 4])) * D)
    kind = rng.choice(["tiny", "short", "tile_edge", "medium", "window"], p=[0.1, 0.3, 0.3, 0.22, 0.08])
    L = {"tiny": lambda: int(rng.integers(2, 8)), "short": lambda: int(rng.integers(8, 62)),
         "tile_edge": lambda: int(rng.choice([30, 62, 94, 126, 190, 254])) + int(rng.integers(-1, 2)),
         "medium": lambda: int(rng.integers(62, 330)), "window": lambda: int(rng.integers(1023, 1060))}[kind]()
    mode = str(rng.choice(["substitutions", "substitutions", "indels", "sliding"]))
    return dict(cfg=cfg, L=L, mode=mode, mirror=bool(rng.random() < 0.7), share=bool(rng.random() < 0.7), seed=int(rng.integers(1 << 30)))


@pytest.mark.parametrize("case", range(max(1, CASES // 3)))
def test_random_tranception_shape_vs_oracle(lib, case):
    """The same for Tranception (model_pytorch.py:878-928, scoring_utils.py:77-203): heads x 64, depth, protein length up to and
    beyond the 1 024-token context, substitutions (prefix-shared or every sequence in full) / indel libraries / sliding windows,
    one or both reading directions, with the wild type among the rows or not."""
    import pandas as pd
    import torch
    from oracle import tranception_oracle as to
    from proteingym_amd import tranception as ptr
    rng = np.random.default_rng([SEED, 1000 + case])
    c = _draw_tranception(rng)
    cfg, L = c["cfg"], c["L"]
    blob = synthetic.random_tranception_weights(cfg, seed=c["seed"] % 100000)
    ocfg, W = to.from_arrays(arrays=synthetic.tranception_blob_to_arrays(cfg, blob), **cfg)
    wt = synthetic.random_sequence(rng, L)
    n = int(rng.integers(1, 24))
    if c["mode"] == "indels":
        _, seqs = synthetic.random_indel_library(seed=c["seed"] % 9973, L=L, n=n, max_edit=min(3, max(1, L - 5)))
        seqs = list(dict.fromkeys(s for s in seqs if len(s) >= 2 and s != wt))
        if rng.random() < 0.5:
            seqs.insert(int(rng.integers(0, len(seqs) + 1)), wt)
        df = pd.DataFrame({"mutant": seqs, "mutated_sequence": seqs})        # the indel files carry the sequence in both (model_pytorch.py:890-893)
    else:
        muts = list(dict.fromkeys(_library(rng, wt, 1, n)))
        df = pd.DataFrame({"mutant": muts, "mutated_sequence": [ptr.get_mutated_sequence(wt, m) for m in muts]})
        df = df[df["mutated_sequence"] != wt].drop_duplicates("mutated_sequence")      # (a multi-mutant may undo itself)
        if rng.random() < 0.3:
            df = pd.concat([df, pd.DataFrame({"mutant": [wt[0] + "1" + wt[0]], "mutated_sequence": [wt]})], ignore_index=True)
    window = "sliding" if c["mode"] == "sliding" else "optimal"
    what = f"case {case}: Tranception {cfg['layers']}x{cfg['embed_dim']} ({cfg['heads']} heads) ffn {cfg['ffn_dim']} L={L} {c['mode']} " \
           f"mirror={c['mirror']} share_prefix={c['share']} rows={len(df)}"
    if len(df) == 0:
        pytest.skip(what + ": empty library")
    model = ptr.TranceptionModel(cfg, blob, device=0, scoring_window=window)
    model.share_prefix = c["share"]
    try:
        with torch.no_grad():
            want = to.score_mutants(ocfg, W, df, wt, scoring_mirror=c["mirror"], scoring_window=window, indel_mode=c["mode"] == "indels")
        have = model.score_mutants(DMS_data=df, target_seq=wt, scoring_mirror=c["mirror"], indel_mode=c["mode"] == "indels")
        assert sorted(have.columns) == sorted(want.columns), (what, list(have.columns), list(want.columns))
        key = "mutated_sequence"
        h = have[have[key].notna()].set_index(key)
        w = want[want[key].notna()].set_index(key)
        assert sorted(h.index) == sorted(w.index), what
        cols = [x for x in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score") if x in w.columns]
        worst = max(float(np.abs(h.loc[w.index, x].to_numpy(dtype=np.float64) - w[x].to_numpy(dtype=np.float64)).max()) for x in cols)
        assert len(have) == len(want), what                               # the zero row of the wild type, when it is among the inputs
        print(what, f"avg scores max|err| {worst:.2e}")
        assert worst < TOL, what
    finally:
        model.close()


@pytest.mark.parametrize("case", range(max(1, CASES // 4)))
def test_random_msa_transformer_grid_vs_oracle(lib, case):
    """The same for the MSA Transformer (esm/model/msa_transformer.py:146-205, esm/axial_attention.py): heads x 64, depth, R rows x
    C columns on both sides of every tile edge (one row -- an alignment of the query alone --, R < 32, C % 32, the split of the tied
    scores over the rows): the whole grid's log-probabilities and the masked-marginals rows (compute_fitness.py:380-394; the last
    layer runs for the kept column only) against the oracle."""
    import torch
    from oracle import msa_transformer_oracle as mo
    from proteingym_amd import msa_transformer as pmsa
    rng = np.random.default_rng([SEED, 2000 + case])
    heads = int(rng.choice([1, 2, 4, 12]))
    cfg = dict(arch=4, layers=int(rng.integers(1, 4)), embed_dim=64 * heads, heads=heads, ffn_dim=int(rng.choice([2, 4])) * 64 * heads,
               max_positions=1024, embed_positions_msa=True)
    R = int(rng.choice([1, 2, int(rng.integers(3, 31)), int(rng.choice([31, 32, 33])), int(rng.integers(34, 130))]))
    C = int(rng.choice([int(rng.integers(3, 31)), int(rng.choice([31, 32, 33, 63, 64, 65])), int(rng.integers(66, 200))]))
    arrays = synthetic.random_msa_transformer_arrays(cfg, seed=int(rng.integers(100000)))
    blob = pmsa.pack_state_dict(cfg, arrays)
    tok = rng.integers(4, 30, size=(R, C)).astype(np.int64)
    tok[:, 0] = 0                                                        # <cls>
    tok[1:][rng.random((R - 1, C)) < 0.05] = 30                          # gaps in the members
    tok[:, 0] = 0
    what = f"case {case}: MSA Transformer {cfg['layers']}x{cfg['embed_dim']} ({heads} heads) ffn {cfg['ffn_dim']} grid {R} x {C}"
    m = pmsa.MsaTransformerModel(cfg, blob, max_rows=max(2048, ((R + 31) // 32 * 32) * ((C + 31) // 32 * 32)))
    try:
        ocfg, W = mo.from_arrays(arrays=arrays, **cfg)
        with torch.no_grad():
            ref = torch.log_softmax(mo.forward_logits(ocfg, W, tok), -1).numpy()
        lp = m.token_logprobs(tok)
        err = float(np.abs(lp - ref).max())
        positions = sorted({int(p) for p in rng.integers(1, C, size=min(4, C - 1))})
        rows = m.masked_logprobs(tok, positions, seq_len=C - 1)
        table = mo.masked_marginals_table(ocfg, W, tok, C - 1, positions=positions)
        err_m = float(np.abs(rows - table[positions]).max())
        print(what, f"grid max|err| {err:.2e}, masked rows max|err| {err_m:.2e}")
        assert err < TOL and err_m < TOL, what
    finally:
        m.close()


@pytest.mark.parametrize("case", range(max(1, CASES // 9)))
def test_random_assay_tables_through_the_runner(lib, tmp_path, case):
    """run_benchmark over a drawn table of assays (lengths 8 ... 420, rows 1 ... 300, offsets, a second checkpoint or not) in its three
    work-unit forms -- whole assays one at a time, short assays grouped behind key masks, chunks of masked positions -- writes the scores of
    ``Assay.run()`` on each assay alone, bit for bit, and the same files in all three."""
    import pandas as pd
    from proteingym_amd import run_benchmark as rb
    rng = np.random.default_rng([SEED, 3000 + case])
    arch = synthetic.ESM2_650M if rng.random() < 0.4 else synthetic.ESM1V_650M
    heads = int(rng.choice([2, 4]))
    cfg = dict(arch, layers=int(rng.integers(1, 3)), embed_dim=64 * heads, heads=heads, ffn_dim=256 * heads)     # (an ESM2 file says 4 x embed_dim)
    stems = (["esm2_t_a"] if arch is synthetic.ESM2_650M else ["esm1v_a", "esm1v_b"][: int(rng.integers(1, 3))])
    for k, stem in enumerate(stems):
        synthetic.save_fair_esm_checkpoint(str(tmp_path / f"{stem}.pt"), cfg, synthetic.random_weights(cfg, seed=int(rng.integers(100000)), embed_std=0.3))
    rows, assays = [], {}
    for k in range(int(rng.integers(2, 7))):
        L = int(rng.choice([int(rng.integers(8, 60)), int(rng.integers(60, 200)), int(rng.integers(200, 420))]))
        offset = int(rng.choice([1, 1, 5, 120]))
        seq = synthetic.random_sequence(rng, L)
        muts = _library(rng, seq, offset, int(rng.integers(1, 300)))
        pd.DataFrame({"mutant": muts, "DMS_score": rng.standard_normal(len(muts))}).to_csv(tmp_path / f"A{k}.csv", index=False)
        rows.append({"DMS_id": f"A{k}", "DMS_filename": f"A{k}.csv", "target_seq": seq, "DMS_total_number_mutants": len(muts), "start_idx": offset})
        assays[f"A{k}"] = (seq, muts, offset)
    pd.DataFrame(rows).to_csv(tmp_path / "map.csv", index=False)
    common = ["--model-location"] + [str(tmp_path / f"{s}.pt") for s in stems] + ["--model_type", "ESM2" if arch is synthetic.ESM2_650M else "ESM1v",
              "--dms_mapping", str(tmp_path / "map.csv"), "--dms-input", str(tmp_path)]
    modes = {"one": ["--batch-short-tokens", "0"], "groups": ["--batch-short-tokens", "250", "--batch-short-rows", "400"],
             "positions": ["--shard", "positions", "--chunk-forwards", str(int(rng.integers(3, 40)))]}
    for name, extra in modes.items():
        rb.main(rb.create_parser().parse_args(common + ["--dms-output", str(tmp_path / name)] + extra))
    models = [pesm.load_model_and_alphabet(str(tmp_path / f"{s}.pt"))[0] for s in stems]
    try:
        for a, (seq, muts, offset) in assays.items():
            ref = open(tmp_path / "one" / f"{a}.csv").read()
            assert open(tmp_path / "groups" / f"{a}.csv").read() == ref and open(tmp_path / "positions" / f"{a}.csv").read() == ref, (case, a)
            got = pd.read_csv(tmp_path / "one" / f"{a}.csv", float_precision="round_trip")
            for s, m in zip(stems, models):
                alone = pesm.Assay(m, seq, muts, offset_idx=offset)
                try:
                    assert np.array_equal(got[s].to_numpy(), alone.run()), (case, a, s)
                finally:
                    alone.close()
    finally:
        for m in models:
            m.close()


@pytest.mark.parametrize("case", range(max(1, CASES // 6)))
def test_random_tranception_retrieval_vs_oracle(lib, tmp_path, case):
    """Tranception with inference-time retrieval (model_pytorch.py:806-830, msa_utils.py:63-138) on a drawn alignment: members 5 - 60 %
    away from the query with gaps, covering residues [MSA_start, MSA_end) of a protein that may be longer than the alignment (and, now and
    then, than the context: every scoring window overlaps the prior differently, in both reading directions), a drawn fusion weight."""
    import pandas as pd
    import torch
    from oracle import tranception_oracle as to
    from proteingym_amd import tranception as ptr
    rng = np.random.default_rng([SEED, 4000 + case])
    heads = int(rng.choice([4, 8]))
    cfg = dict(synthetic.TRANCEPTION_L, layers=int(rng.integers(1, 3)), embed_dim=64 * heads, heads=heads, ffn_dim=128 * heads)
    L = int(rng.choice([int(rng.integers(12, 60)), int(rng.integers(60, 260)), int(rng.integers(1023, 1050))], p=[0.4, 0.5, 0.1]))
    wt = synthetic.random_sequence(rng, L)
    ms = int(rng.integers(0, max(1, L // 3)))
    me = int(rng.integers(min(L, ms + 6), L + 1)) if rng.random() < 0.7 else L
    span = np.array(list(wt[ms:me]))
    lines = [f">query/{ms + 1}-{me}", "".join(span)]
    for k in range(int(rng.integers(3, 50))):
        row = span.copy()
        flip = rng.random(row.size) < rng.uniform(0.05, 0.6)
        row[flip] = rng.choice(list(synthetic.AA), size=int(flip.sum()))
        row[rng.random(row.size) < 0.05] = "-"
        lines += [f">member{k}/{ms + 1}-{me}", "".join(row)]
    a2m = tmp_path / "drawn.a2m"
    a2m.write_text("\n".join(lines) + "\n")
    weight = float(rng.choice([0.6, 0.3, 0.9]))
    blob = synthetic.random_tranception_weights(cfg, seed=int(rng.integers(100000)))
    ocfg, W = to.from_arrays(arrays=synthetic.tranception_blob_to_arrays(cfg, blob), **cfg)
    muts = list(dict.fromkeys(_library(rng, wt, 1, int(rng.integers(2, 20)))))
    df = pd.DataFrame({"mutant": muts, "mutated_sequence": [ptr.get_mutated_sequence(wt, m) for m in muts]})
    df = df[df["mutated_sequence"] != wt].drop_duplicates("mutated_sequence")
    what = f"case {case}: Tranception + retrieval {cfg['layers']}x{cfg['embed_dim']} L={L} alignment [{ms}, {me}) weight {weight} rows={len(df)}"
    if len(df) == 0:
        pytest.skip(what + ": empty library")
    prior = to.get_msa_prior(str(a2m), ms, me, L)
    retrieval = dict(log_prior=torch.log(torch.tensor(prior).float()).numpy(), MSA_start=ms, MSA_end=me, weight=weight)
    model = ptr.TranceptionModel(cfg, blob, device=0)
    try:
        model.retrieval = ptr.build_retrieval(dict(MSA_filename=str(a2m), MSA_start=ms, MSA_end=me, full_protein_length=L,
                                                   retrieval_inference_weight=weight))
        assert np.abs(np.exp(model.retrieval["log_prior"]) - prior).max() < 1e-6, what
        with torch.no_grad():
            want = to.score_mutants(ocfg, W, df, wt, retrieval=retrieval)
        have = model.score_mutants(DMS_data=df, target_seq=wt)
        h, w = have.set_index("mutated_sequence"), want.set_index("mutated_sequence")
        worst = max(float(np.abs(h.loc[w.index, x].to_numpy(dtype=np.float64) - w[x].to_numpy(dtype=np.float64)).max())
                    for x in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"))
        print(what, f"avg scores max|err| {worst:.2e}")
        assert worst < TOL, what
    finally:
        model.close()
