"""Golden fixtures for ESM2 checkpoints whose head_dim is below 64 (the 8M / 35M / 150M members of the
family use 16 / 24 / 32), produced by the UNMODIFIED reference on CPU (oracle/ref_harness.py):

    python tests/golden/make_golden_small_heads.py

  esm2_toy_h16.pt  2 layers, D=128, 8 heads  (head_dim 16)
  esm2_toy_h24.pt  2 layers, D= 96, 4 heads  (head_dim 24; D is not a multiple of 64 -> fp32 mode only)
  esm2_toy_h32.pt  2 layers, D= 64, 2 heads  (head_dim 32)
  golden_esm_small_heads.npz:  <ckpt>/wt_logprobs, <ckpt>/mm_table, <ckpt>/pad_tokens, <ckpt>/pad_logprobs
                               (as in make_golden.py) and cli/<ckpt> = the reference CLI's score column on
                               TOY_DMS.csv (masked-marginals).
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


def main():
    g = np.load(os.path.join(HERE, "golden_esm.npz"))
    seq = str(g["seq"])
    ck = {
        "esm2_toy_h16": rh.make_esm2_checkpoint(os.path.join(HERE, "esm2_toy_h16.pt"), 2, 128, 8, seed=16, embed_std=0.25),
        "esm2_toy_h24": rh.make_esm2_checkpoint(os.path.join(HERE, "esm2_toy_h24.pt"), 2, 96, 4, seed=24, embed_std=0.25),
        "esm2_toy_h32": rh.make_esm2_checkpoint(os.path.join(HERE, "esm2_toy_h32.pt"), 2, 64, 2, seed=32, embed_std=0.25),
    }
    out = {}
    rh.load_reference()
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
            _, _, pt = alphabet.get_batch_converter()([("a", seq), ("b", seq[:41])])
            out[f"{name}/pad_tokens"] = pt.numpy()
            out[f"{name}/pad_logprobs"] = torch.log_softmax(model(pt)["logits"], dim=-1).numpy()
    with tempfile.TemporaryDirectory() as d:
        for nm in ck:
            rh.run_reference_cli(["--model-location", ck[nm], "--model_type", "ESM2",
                                  "--dms-input", os.path.join(HERE, "TOY_DMS.csv"), "--dms-output", os.path.join(d, "o_" + nm),
                                  "--target_seq", seq, "--scoring-strategy", "masked-marginals", "--nogpu"])
            out[f"cli/{nm}"] = pd.read_csv(os.path.join(d, "o_" + nm, "TOY_DMS.csv"))[nm].to_numpy()
    np.savez_compressed(os.path.join(HERE, "golden_esm_small_heads.npz"), **out)
    print("wrote golden_esm_small_heads.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
