"""Golden vectors for two more modes of the Tranception scorer, produced by the REFERENCE on CPU
(oracle/ref_harness.py): indel scoring (variable-length mutated sequences, --indel_mode) and the
'sliding' scoring window on a protein longer than the context.

    python tests/golden/make_golden_tranception_modes.py   ->  TOY_TRANCEPTION_INDEL_DMS.csv, golden_tranception_modes.npz
"""
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_harness as rh  # noqa: E402

AA = "ACDEFGHIKLMNPQRSTVWY"


def main():
    rng = np.random.default_rng(99)
    g = np.load(os.path.join(HERE, "golden_tranception.npz"))
    seq, seq_long = str(g["seq"]), str(g["seq_long"])
    ck = os.path.join(HERE, "Tranception_toy")
    # indels: deletions, insertions and the wild type itself
    rows = [seq]
    for _ in range(14):
        s = list(seq)
        p = int(rng.integers(1, len(s) - 6))
        if rng.random() < 0.5:
            del s[p:p + int(rng.integers(1, 5))]
        else:
            s[p:p] = list(rng.choice(list(AA), size=int(rng.integers(1, 5))))
        rows.append("".join(s))
    score = rng.standard_normal(len(rows))
    indel = pd.DataFrame({"mutant": rows, "mutated_sequence": rows, "DMS_score": score})
    indel.to_csv(os.path.join(HERE, "TOY_TRANCEPTION_INDEL_DMS.csv"), index=False)
    out = {}
    model, _ = rh.reference_tranception_model(ck)
    r = model.score_mutants(DMS_data=indel, target_seq=seq, scoring_mirror=True, batch_size_inference=4, num_workers=0, indel_mode=True)
    # reference quirk (model_pytorch.py:915-924): in indel mode the zero-score wild-type row is appended with the
    # sequence in column 'mutant', so its 'mutated_sequence' is NaN
    out["indel/columns"] = np.array(list(r.columns))
    wt = r[r["mutated_sequence"].isna()]
    assert len(wt) == 1 and wt["mutant"].iloc[0] == seq and float(wt["avg_score"].iloc[0]) == 0.0
    r = pd.merge(indel[["mutated_sequence"]].iloc[1:], r, on="mutated_sequence", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        out[f"indel/{c}"] = r[c].to_numpy()
    # sliding windows on the 1100-residue protein
    model_s, _ = rh.reference_tranception_model(ck, scoring_window="sliding")
    dms_long = pd.read_csv(os.path.join(HERE, "TOY_TRANCEPTION_LONG_DMS.csv"))
    r = model_s.score_mutants(DMS_data=dms_long, target_seq=seq_long, scoring_mirror=True, batch_size_inference=4, num_workers=0, indel_mode=False)
    r = pd.merge(dms_long[["mutated_sequence"]], r, on="mutated_sequence", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        out[f"sliding/{c}"] = r[c].to_numpy()
    np.savez_compressed(os.path.join(HERE, "golden_tranception_modes.npz"), **out)
    print({k: v[:3] for k, v in out.items()})


if __name__ == "__main__":
    main()
