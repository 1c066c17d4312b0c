"""Golden fixtures for an ESM2 checkpoint with head_dim 128 -- the shape class of ESM2-15B (esm2_t48_15B_UR50D: 48 x 5120,
40 heads; /root/reference/proteingym/baselines/esm/esm/pretrained.py:387-394, config.json "ESM2_15B") -- produced by the
UNMODIFIED reference on CPU (oracle/ref_harness.py):

    python tests/golden/make_golden_h128.py

  esm2_toy_h128.pt   2 layers, D=256, 2 heads (head_dim 128, rotary over 128 dims)
  golden_esm_h128.npz:  wt_logprobs, mm_table, pad_tokens, pad_logprobs (as in make_golden.py) and cli = the reference
                        CLI's score column on TOY_DMS.csv (masked-marginals)
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
    path = rh.make_esm2_checkpoint(os.path.join(HERE, "esm2_toy_h128.pt"), 2, 256, 2, seed=128, embed_std=0.25)
    out = {}
    rh.load_reference()
    model, alphabet = rh.reference_model(path)
    assert model.layers[0].self_attn.head_dim == 128
    _, _, toks = alphabet.get_batch_converter()([("protein1", seq)])
    with torch.no_grad():
        out["wt_logprobs"] = torch.log_softmax(model(toks)["logits"], dim=-1)[0].numpy()
        rows = []
        for i in range(toks.size(1)):                      # compute_fitness.py:489-503
            t = toks.clone()
            t[0, i] = alphabet.mask_idx
            rows.append(torch.log_softmax(model(t)["logits"], dim=-1)[:, i])
        out["mm_table"] = torch.cat(rows, dim=0).numpy()
        _, _, pt = alphabet.get_batch_converter()([("a", seq), ("b", seq[:41])])
        out["pad_tokens"] = pt.numpy()
        out["pad_logprobs"] = torch.log_softmax(model(pt)["logits"], dim=-1).numpy()
    with tempfile.TemporaryDirectory() as d:
        rh.run_reference_cli(["--model-location", path, "--model_type", "ESM2", "--dms-input", os.path.join(HERE, "TOY_DMS.csv"),
                              "--dms-output", os.path.join(d, "o"), "--target_seq", seq, "--scoring-strategy", "masked-marginals", "--nogpu"])
        out["cli"] = pd.read_csv(os.path.join(d, "o", "TOY_DMS.csv"))["esm2_toy_h128"].to_numpy()
    np.savez_compressed(os.path.join(HERE, "golden_esm_h128.npz"), **out)
    print("wrote golden_esm_h128.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
