"""Golden fixtures for the MSA Transformer path, produced by the UNMODIFIED reference on CPU
(oracle/ref_harness.py): model outputs and the reference CLI's score columns.

    python tests/golden/make_golden_msa_transformer.py

  msa_toy.pt                  2 layers, D=128, 2 heads, F=256, random weights (reference constructor), file
                              layout of esm_msa1b_t12_100M_UR50S (swapped row/column names, encoder. prefixes)
  TOY_MSA_DMS.csv             mutants inside residues 6..65 of the 70-residue toy protein (TOY_MSA_GAPPY.a2m span)
  TOY_MSA_MAPPING.csv         one-row reference file (DMS_id, target_seq, MSA_filename, MSA_start/end, weight_file_name)
  TOY_MSA_LONG.a2m / _weights.npy / TOY_MSA_LONG_DMS.csv   8 sequences x 1100 columns (optimal 1024 window)
  golden_msa_transformer.npz:
     logits_tokens, logits            reference model on a 5 x 22 token grid
     sampled/seed{1,2}                the token grids the reference sampled (sequence-reweighting, 12 rows)
     mm_table/seed1                   masked-marginals table rebuilt with the reference model on sampled/seed1
     cli/<col>                        CLI score columns: msa_toy_seed1, msa_toy_seed2, msa_toy_ensemble
     cli_long/msa_toy_seed1           same for the 1100-column alignment
"""
import os
import sys
import tempfile

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import ref_harness as rh, tranception_oracle as to  # noqa: E402

AA = "ACDEFGHIKLMNPQRSTVWY"


def synth_dms(rng, seq, lo, hi, n_single, n_multi):
    """mutants with 1-indexed positions in [lo, hi] of seq"""
    rows = []
    for _ in range(n_single):
        p = int(rng.integers(lo - 1, hi))
        rows.append(f"{seq[p]}{p + 1}{rng.choice([a for a in AA if a != seq[p]])}")
    for _ in range(n_multi):
        ps = sorted(rng.choice(np.arange(lo - 1, hi), size=int(rng.integers(2, 5)), replace=False))
        rows.append(":".join(f"{seq[p]}{p + 1}{rng.choice([a for a in AA if a != seq[p]])}" for p in ps))
    score = rng.standard_normal(len(rows))
    return pd.DataFrame({"mutant": rows, "DMS_score": score, "DMS_score_bin": (score > 0).astype(int)})


def main():
    rng = np.random.default_rng(31337)
    cf = rh.load_reference()
    ck = rh.make_msa_transformer_checkpoint(os.path.join(HERE, "msa_toy.pt"), 2, 128, 256, 2, seed=12)
    model, alphabet = rh.reference_model(ck)
    conv = alphabet.get_batch_converter()
    out = {}
    g = np.load(os.path.join(HERE, "golden_esm.npz"))
    seq = str(g["seq"])
    gw = np.load(os.path.join(HERE, "golden_msa_weights.npz"))
    ms, me = int(gw["msa_start_end"][0]) + 1, int(gw["msa_start_end"][1])           # 1-indexed inclusive: 6..65

    # 1. raw model on a small grid
    msa = [(f"s{i}", "".join(rng.choice(list(AA + "-"), size=21))) for i in range(5)]
    _, _, toks = conv([msa])
    with torch.no_grad():
        out["logits"] = model(toks)["logits"][0].numpy()
    out["logits_tokens"] = toks[0].numpy()

    # 2. CLI, index mode (crops target_seq to the MSA span, offset = MSA_start)
    dms = synth_dms(rng, seq, ms, me, 40, 15)
    dms.to_csv(os.path.join(HERE, "TOY_MSA_DMS.csv"), index=False)
    pd.DataFrame([{"DMS_id": "TOY_MSA_DMS", "DMS_filename": "TOY_MSA_DMS.csv", "target_seq": seq,
                   "MSA_filename": "TOY_MSA_GAPPY.a2m", "MSA_start": ms, "MSA_end": me,
                   "weight_file_name": "TOY_MSA_GAPPY_weights.npy"}]).to_csv(os.path.join(HERE, "TOY_MSA_MAPPING.csv"), index=False)
    with tempfile.TemporaryDirectory() as d:
        rh.run_reference_cli(["--model-location", ck, "--model_type", "MSA_transformer", "--dms_index", "0",
                              "--dms_mapping", os.path.join(HERE, "TOY_MSA_MAPPING.csv"), "--dms-input", HERE,
                              "--dms-output", os.path.join(d, "o"), "--scoring-strategy", "masked-marginals",
                              "--scoring-window", "optimal", "--msa-path", HERE, "--msa-weights-folder", HERE,
                              "--msa-samples", "12", "--seeds", "1", "2", "--nogpu"])
        df = pd.read_csv(os.path.join(d, "o", "TOY_MSA_DMS.csv"))
        out["cli/columns"] = np.array(list(df.columns))
        for c in ("msa_toy_seed1", "msa_toy_seed2", "msa_toy_ensemble"):
            out[f"cli/{c}"] = df[c].to_numpy()
    # the sampled grids and one table, through the reference functions directly
    pm = cf.process_msa(filename=os.path.join(HERE, "TOY_MSA_GAPPY.a2m"), weight_filename=os.path.join(HERE, "TOY_MSA_GAPPY_weights.npy"),
                        filter_msa=False, path_to_hhfilter=None)
    for seed in (1, 2):
        data = [cf.sample_msa(sampling_strategy="sequence-reweighting", filename=None, nseq=12, weight_filename=None,
                              processed_msa=pm, random_seed=seed)]
        _, _, bt = conv(data)
        out[f"sampled/seed{seed}"] = bt[0].numpy()
    bt = torch.as_tensor(out["sampled/seed1"])[None]
    rows = []
    with torch.no_grad():
        for i in range(bt.size(2)):
            t = bt.clone()
            t[0, 0, i] = alphabet.mask_idx
            rows.append(torch.log_softmax(model(t)["logits"], dim=-1)[:, 0, i])
    out["mm_table/seed1"] = torch.cat(rows, 0).numpy()

    # 3. long alignment: 1100 columns -> optimal 1024-column window per masked position
    seq_long = str(g["seq_long"])
    lines = [f">LONG/1-{len(seq_long)}", seq_long]
    for i in range(7):
        s = list(seq_long)
        for p in rng.choice(len(s), size=150, replace=False):
            s[p] = rng.choice(list(AA + "-"))
        lines += [f">l{i}/1-{len(seq_long)}", "".join(s)]
    a2m = os.path.join(HERE, "TOY_MSA_LONG.a2m")
    open(a2m, "w").write("\n".join(lines) + "\n")
    w = to.eve_sequence_weights(a2m)
    np.save(os.path.join(HERE, "TOY_MSA_LONG_weights.npy"), np.array(list(w.values())))
    dms_long = synth_dms(rng, seq_long, 1, len(seq_long), 16, 6)
    dms_long.to_csv(os.path.join(HERE, "TOY_MSA_LONG_DMS.csv"), index=False)
    with tempfile.TemporaryDirectory() as d:
        rh.run_reference_cli(["--model-location", ck, "--model_type", "MSA_transformer",
                              "--dms-input", os.path.join(HERE, "TOY_MSA_LONG_DMS.csv"), "--dms-output", os.path.join(d, "o"),
                              "--target_seq", seq_long, "--scoring-strategy", "masked-marginals", "--scoring-window", "optimal",
                              "--msa-path", a2m, "--msa-weights-folder", HERE, "--weight_file_name", "TOY_MSA_LONG_weights.npy",
                              "--msa-samples", "6", "--seeds", "1", "--nogpu"])
        out["cli_long/msa_toy_seed1"] = pd.read_csv(os.path.join(d, "o", "TOY_MSA_LONG_DMS.csv"))["msa_toy_seed1"].to_numpy()
    np.savez_compressed(os.path.join(HERE, "golden_msa_transformer.npz"), **out)
    print("wrote golden_msa_transformer.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
