"""Golden fixtures for pseudo-perplexity scoring of VARIABLE-LENGTH (indel) mutants -- BASELINE config 5.

Runs the UNMODIFIED reference CLI (/root/reference/proteingym/baselines/esm/compute_fitness.py, main():282-543 with
``--scoring-strategy pseudo-ppl``: compute_pppl :258-279, driver :515-529) through oracle/ref_harness.py on a toy
indel file whose ``mutated_sequence`` column holds insertions, deletions, mixed edits, the wild type and three
degenerate lengths (2, 3 and 4 residues: zero, one and two scored terms).  Run once in the build container:

    python tests/golden/make_golden_pppl_indels.py

Outputs (committed):
  TOY_INDELS.csv                  mutant (label), mutated_sequence, DMS_score, DMS_score_bin
  golden_pppl_indels.npz
      cli/<ckpt>                  the score column the reference CLI wrote (python-float sums of f32 terms)
      cli/columns                 the CSV header the reference CLI wrote
      terms/<ckpt>/<row>          per-position terms log p(sequence[i] | token i masked), i = 1..len-2, taken from
                                  the reference MODEL with the reference's loop (same off-by-one): what the 1e-4
                                  per-term bar is checked against
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


def make_variants(rng, seq):
    L = len(seq)
    rows = [("WT", seq)]
    for k in (1, 2, 3, 7):                                   # insertions
        p = int(rng.integers(1, L - 1))
        ins = "".join(rng.choice(list(AA), size=k))
        rows.append((f"ins{k}@{p}", seq[:p] + ins + seq[p:]))
    for k in (1, 2, 3, 9):                                   # deletions
        p = int(rng.integers(1, L - k - 1))
        rows.append((f"del{k}@{p}", seq[:p] + seq[p + k:]))
    p, q = 5, 40                                             # deletion + insertion in one variant
    rows.append(("delins", seq[:p] + seq[p + 2:q] + "WWK" + seq[q:]))
    rows.append(("ins_front", "MK" + seq))
    rows.append(("del_tail", seq[:-4]))
    rows.append(("half", seq[: L // 2]))                     # a much shorter member of the library
    rows.append(("len4", seq[:4]))                           # two terms
    rows.append(("len3", seq[:3]))                           # one term
    rows.append(("len2", seq[:2]))                           # range(1, 1): no term, score 0
    return rows


def main():
    rng = np.random.default_rng(20260925)
    g = np.load(os.path.join(HERE, "golden_esm.npz"))
    seq = str(g["seq"])
    rows = make_variants(rng, seq)
    score = rng.standard_normal(len(rows))
    df = pd.DataFrame({"mutant": [r[0] for r in rows], "mutated_sequence": [r[1] for r in rows],
                       "DMS_score": score, "DMS_score_bin": (score > 0).astype(int)})
    csv = os.path.join(HERE, "TOY_INDELS.csv")
    df.to_csv(csv, index=False)
    out = {}
    rh.load_reference()
    ckpts = {"esm2_toy": "ESM2", "esm1v_toy_1": "ESM1v"}
    with tempfile.TemporaryDirectory() as d:
        for name, mtype in ckpts.items():
            path = os.path.join(HERE, name + ".pt")
            rh.run_reference_cli(["--model-location", path, "--model_type", mtype, "--dms-input", csv,
                                  "--dms-output", os.path.join(d, "o_" + name), "--target_seq", seq,
                                  "--scoring-strategy", "pseudo-ppl", "--nogpu"])
            got = pd.read_csv(os.path.join(d, "o_" + name, "TOY_INDELS.csv"))
            out[f"cli/{name}"] = got[name].to_numpy()
            out[f"cli/{name}/columns"] = np.array(list(got.columns))
            model, alphabet = rh.reference_model(path)
            for r, (_, s) in enumerate(rows):                # the reference's loop on the reference model
                _, _, toks = alphabet.get_batch_converter()([("protein1", s)])
                terms = []
                for i in range(1, len(s) - 1):
                    t = toks.clone()
                    t[0, i] = alphabet.mask_idx
                    with torch.no_grad():
                        lp = torch.log_softmax(model(t)["logits"], dim=-1)
                    terms.append(lp[0, i, alphabet.get_idx(s[i])].item())
                out[f"terms/{name}/{r}"] = np.array(terms, dtype=np.float64)
                assert abs(sum(terms) - out[f"cli/{name}"][r]) < 1e-9, (name, r)
    np.savez_compressed(os.path.join(HERE, "golden_pppl_indels.npz"), **out)
    print("wrote golden_pppl_indels.npz:", len(out), "arrays,", len(rows), "variants, lengths",
          sorted({len(s) for _, s in rows}))


if __name__ == "__main__":
    main()
