"""Generates the golden fixtures in this directory by running the UNMODIFIED reference
(/root/reference, via oracle/ref_harness.py) on CPU.  Run once in the build container:

    python tests/golden/make_golden.py

Outputs (committed; they travel to the GPU box where /root/reference does not exist):
  esm1v_toy_{1,2}.pt, esm1b_toy_lnb.pt, esm2_toy.pt   random-weight checkpoints built with the
        reference constructors (esm/model/esm1.py:49-105, esm/model/esm2.py:40-74) in fair-esm
        v1 / v2 file layout, D=128, H=2 (head_dim 64), F=256 (ESM2: 512), 2 layers
  TOY_DMS.csv / TOY_LONG_DMS.csv                       synthetic DMS files (singles + multiples)
  golden_esm.npz                                        reference outputs:
        <ckpt>/wt_logprobs      log_softmax(model(tokens)["logits"])       [L+2,33]
        <ckpt>/mm_table         masked-marginals token_probs (compute_fitness.py:486-504)
        <ckpt>/pad_logprobs     a 2-sequence padded batch through the reference model
        cli/<column>            score columns written by the reference CLI (main(), :282-543)
        cli_long/<column>       same for a 1100-residue protein (optimal 1024 window, :492-495)
        cli_wt/<column>         wt-marginals strategy (:433-485)
        cli_pppl/<column>       pseudo-ppl strategy on 6 mutants (:515-529, 258-279)
"""
import os
import sys
import tempfile

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import ref_harness as rh  # noqa: E402

AA = "ACDEFGHIKLMNPQRSTVWY"


def synth_seq(rng, L):
    return "".join(rng.choice(list(AA), size=L))


def synth_dms(rng, seq, n_single, n_multi, offset=1):
    rows = []
    L = len(seq)
    for _ in range(n_single):
        p = int(rng.integers(0, L))
        mt = rng.choice([a for a in AA if a != seq[p]])
        rows.append(f"{seq[p]}{p + offset}{mt}")
    for _ in range(n_multi):
        k = int(rng.integers(2, 6))
        ps = sorted(rng.choice(L, size=k, replace=False))
        rows.append(":".join(f"{seq[p]}{p + offset}{rng.choice([a for a in AA if a != seq[p]])}" for p in ps))
    score = rng.standard_normal(len(rows))
    return pd.DataFrame({"mutant": rows, "DMS_score": score, "DMS_score_bin": (score > 0).astype(int)})


def main():
    rng = np.random.default_rng(20250925)
    out = {}
    ck = {
        "esm1v_toy_1": rh.make_esm1v_checkpoint(os.path.join(HERE, "esm1v_toy_1.pt"), 2, 128, 256, 2, seed=1, embed_std=0.25),
        "esm1v_toy_2": rh.make_esm1v_checkpoint(os.path.join(HERE, "esm1v_toy_2.pt"), 2, 128, 256, 2, seed=2, embed_std=0.25),
        "esm1b_toy_lnb": rh.make_esm1v_checkpoint(os.path.join(HERE, "esm1b_toy_lnb.pt"), 2, 128, 256, 2, seed=3,
                                                  embed_std=0.25, emb_layer_norm_before=True),
        "esm2_toy": rh.make_esm2_checkpoint(os.path.join(HERE, "esm2_toy.pt"), 2, 128, 2, seed=4, embed_std=0.25),
    }
    seq = synth_seq(rng, 70)
    seq_long = synth_seq(rng, 1100)
    out["seq"] = np.array(seq)
    out["seq_long"] = np.array(seq_long)
    dms = synth_dms(rng, seq, 60, 40)
    dms.to_csv(os.path.join(HERE, "TOY_DMS.csv"), index=False)
    dms_long = synth_dms(rng, seq_long, 40, 20)
    dms_long.to_csv(os.path.join(HERE, "TOY_LONG_DMS.csv"), index=False)

    cf = rh.load_reference()
    for name, path in ck.items():
        model, alphabet = rh.reference_model(path)
        _, _, toks = alphabet.get_batch_converter()([("protein1", seq)])
        with torch.no_grad():
            out[f"{name}/wt_logprobs"] = torch.log_softmax(model(toks)["logits"], dim=-1)[0].numpy()
            rows = []
            for i in range(toks.size(1)):                      # compute_fitness.py:489-503
                t = toks.clone()
                t[0, i] = alphabet.mask_idx
                rows.append(torch.log_softmax(model(t)["logits"], dim=-1)[:, i])
            out[f"{name}/mm_table"] = torch.cat(rows, dim=0).numpy()
            # padded batch: two sequences of different length
            _, _, pt = alphabet.get_batch_converter()([("a", seq), ("b", seq[:41])])
            out[f"{name}/pad_tokens"] = pt.numpy()
            out[f"{name}/pad_logprobs"] = torch.log_softmax(model(pt)["logits"], dim=-1).numpy()

    with tempfile.TemporaryDirectory() as d:
        # ESM-1v "ensemble" of the two toy checkpoints through the reference CLI
        rh.run_reference_cli(["--model-location", ck["esm1v_toy_1"], ck["esm1v_toy_2"], "--model_type", "ESM1v",
                              "--dms-input", os.path.join(HERE, "TOY_DMS.csv"), "--dms-output", os.path.join(d, "o1"),
                              "--target_seq", seq, "--scoring-strategy", "masked-marginals",
                              "--scoring-window", "optimal", "--nogpu"])
        df = pd.read_csv(os.path.join(d, "o1", "TOY_DMS.csv"))
        out["cli/columns"] = np.array(list(df.columns))
        for c in ("esm1v_toy_1", "esm1v_toy_2", "Ensemble_ESM1v"):
            out[f"cli/{c}"] = df[c].to_numpy()
        # ESM2 + ESM-1b(lnb) single checkpoints
        for nm, mt in (("esm2_toy", "ESM2"), ("esm1b_toy_lnb", "ESM1b")):
            rh.run_reference_cli(["--model-location", ck[nm], "--model_type", mt,
                                  "--dms-input", os.path.join(HERE, "TOY_DMS.csv"), "--dms-output", os.path.join(d, "o_" + nm),
                                  "--target_seq", seq, "--scoring-strategy", "masked-marginals", "--nogpu"])
            df = pd.read_csv(os.path.join(d, "o_" + nm, "TOY_DMS.csv"))
            out[f"cli/{nm}"] = df[nm].to_numpy()
        # long protein -> optimal windows
        for nm, mt in (("esm1v_toy_1", "ESM1v"), ("esm2_toy", "ESM2")):
            rh.run_reference_cli(["--model-location", ck[nm], "--model_type", mt,
                                  "--dms-input", os.path.join(HERE, "TOY_LONG_DMS.csv"), "--dms-output", os.path.join(d, "ol_" + nm),
                                  "--target_seq", seq_long, "--scoring-strategy", "masked-marginals",
                                  "--scoring-window", "optimal", "--nogpu"])
            df = pd.read_csv(os.path.join(d, "ol_" + nm, "TOY_LONG_DMS.csv"))
            out[f"cli_long/{nm}"] = df[nm].to_numpy()
        # wt-marginals (short + long/overlapping)
        rh.run_reference_cli(["--model-location", ck["esm1b_toy_lnb"], "--model_type", "ESM1b",
                              "--dms-input", os.path.join(HERE, "TOY_DMS.csv"), "--dms-output", os.path.join(d, "ow"),
                              "--target_seq", seq, "--scoring-strategy", "wt-marginals", "--nogpu"])
        out["cli_wt/esm1b_toy_lnb"] = pd.read_csv(os.path.join(d, "ow", "TOY_DMS.csv"))["esm1b_toy_lnb"].to_numpy()
        rh.run_reference_cli(["--model-location", ck["esm1v_toy_1"], "--model_type", "ESM1b",
                              "--dms-input", os.path.join(HERE, "TOY_LONG_DMS.csv"), "--dms-output", os.path.join(d, "owl"),
                              "--target_seq", seq_long, "--scoring-strategy", "wt-marginals",
                              "--scoring-window", "overlapping", "--nogpu"])
        out["cli_wt_long/esm1v_toy_1"] = pd.read_csv(os.path.join(d, "owl", "TOY_LONG_DMS.csv"))["esm1v_toy_1"].to_numpy()
        # pseudo-ppl on the first 6 single mutants
        small = dms.iloc[:6][["mutant", "DMS_score"]]
        small.to_csv(os.path.join(d, "TOY_PPPL.csv"), index=False)
        rh.run_reference_cli(["--model-location", ck["esm2_toy"], "--model_type", "ESM2",
                              "--dms-input", os.path.join(d, "TOY_PPPL.csv"), "--dms-output", os.path.join(d, "op"),
                              "--target_seq", seq, "--scoring-strategy", "pseudo-ppl", "--nogpu"])
        out["cli_pppl/esm2_toy"] = pd.read_csv(os.path.join(d, "op", "TOY_PPPL.csv"))["esm2_toy"].to_numpy()
    np.savez_compressed(os.path.join(HERE, "golden_esm.npz"), **out)
    print("wrote", os.path.join(HERE, "golden_esm.npz"), "with", len(out), "arrays")


if __name__ == "__main__":
    main()
