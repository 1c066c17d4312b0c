"""Golden fixture at ESM2-35M's width (esm2_t12_35M_UR50D: embed_dim 480, 20 heads of 24 -- esm/pretrained.py:355-360, the
registry row /root/reference/config.json:18): the UNMODIFIED reference on CPU (oracle/ref_harness.py) on a 3-layer checkpoint
of that width.

    python tests/golden/make_golden_esm2_35m_width.py

The checkpoint itself is not committed (17 MB): its weights are ``synthetic.random_weights(cfg, seed=35, embed_std=0.15)``, written
to a fair-esm v2 file by ``synthetic.save_fair_esm_checkpoint`` and read from there by the reference's own loader; the GPU test
rebuilds the same file from the same seed (tests/test_gpu_esm.py::test_esm2_35m_width_f16x3_vs_reference).

  golden_esm2_35m_width.npz:  wt_logprobs, mm_table, pad_tokens, pad_logprobs (as in make_golden.py), cli = the reference CLI's
                              masked-marginals score column on TOY_DMS.csv, weights_sha256 of the blob the fixture was made from.
"""
import hashlib
import os
import sys
import tempfile

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import ref_harness as rh  # noqa: E402
from proteingym_amd import synthetic  # noqa: E402

CFG = dict(synthetic.ESM2_35M, layers=3)
SEED, EMBED_STD = 35, 0.15


def main():
    g = np.load(os.path.join(HERE, "golden_esm.npz"))
    seq = str(g["seq"])
    blob = synthetic.random_weights(CFG, seed=SEED, embed_std=EMBED_STD)
    out = {"weights_sha256": np.frombuffer(hashlib.sha256(blob.tobytes()).digest(), dtype=np.uint8)}
    rh.load_reference()
    with tempfile.TemporaryDirectory() as d:
        path = synthetic.save_fair_esm_checkpoint(os.path.join(d, "esm2_toy_35m_width.pt"), CFG, blob)
        model, alphabet = rh.reference_model(path)
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
        rh.run_reference_cli(["--model-location", path, "--model_type", "ESM2",
                              "--dms-input", os.path.join(HERE, "TOY_DMS.csv"), "--dms-output", os.path.join(d, "o"),
                              "--target_seq", seq, "--scoring-strategy", "masked-marginals", "--nogpu"])
        out["cli"] = pd.read_csv(os.path.join(d, "o", "TOY_DMS.csv"))["esm2_toy_35m_width"].to_numpy()
    np.savez_compressed(os.path.join(HERE, "golden_esm2_35m_width.npz"), **out)
    print("wrote golden_esm2_35m_width.npz with", len(out), "arrays; log-prob range",
          float(out["mm_table"].max() - out["mm_table"].min()))


if __name__ == "__main__":
    main()
