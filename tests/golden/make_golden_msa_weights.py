"""Golden vectors for the EVE sequence-weight branch of Tranception retrieval, produced by running
the REFERENCE (tranception/utils/msa_utils.py:194-368 ``MSA_processing`` and :63-138
``get_msa_prior`` with a weights file; model_pytorch.py:806-830 fusion) in this container.

    python tests/golden/make_golden_msa_weights.py

Writes TOY_MSA_GAPPY.a2m (gaps, lower-case inserts, '.', X and mostly-gap rows so that every filter
of MSA_processing fires), TOY_MSA_GAPPY_weights.npy (the reference's computed weights, theta=0.2) and
golden_msa_weights.npz (kept names, weights, prior, retrieval scores on TOY_TRANCEPTION_DMS.csv).
Requires golden_tranception.npz / Tranception_toy (make_golden_tranception.py) to exist.
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
    rh.load_reference_tranception()
    sys.path.insert(0, rh.REF_TRANCEPTION)
    from tranception.utils import msa_utils
    g = np.load(os.path.join(HERE, "golden_tranception.npz"))
    seq = str(g["seq"])
    rng = np.random.default_rng(4242)
    ms, me = 6, 65                                       # 1-indexed inclusive span of the target covered by the MSA
    focus = seq[ms - 1:me]
    lines = [f">TARGET/{ms}-{me}", focus]
    for i in range(90):
        s = list(focus)
        for p in rng.choice(len(s), size=int(rng.integers(1, 28)), replace=False):
            s[p] = rng.choice(list(AA + "--"))
        if i % 11 == 0:
            s[int(rng.integers(len(s)))] = "X"           # indeterminate AA in a focus column: dropped from the weights
        if i % 13 == 0:
            s = ["-"] * 35 + s[35:]                      # > 50 % gaps: dropped by the pre-processing
        if i % 5 == 0:
            s = [c.lower() if rng.random() < 0.2 and c != "-" else c for c in s]   # case is normalised
        if i % 17 == 0:
            s = ["." if c == "-" else c for c in s]
        if i % 9 == 0 and i:
            s = list(lines[-1])                          # exact duplicate of the previous sequence: shared cluster
        lines += [f">seq{i}/1-{len(focus)}", "".join(s)]
    a2m = os.path.join(HERE, "TOY_MSA_GAPPY.a2m")
    open(a2m, "w").write("\n".join(lines) + "\n")
    wfile = os.path.join(HERE, "TOY_MSA_GAPPY_weights.npy")
    if os.path.exists(wfile):
        os.remove(wfile)
    proc = msa_utils.MSA_processing(MSA_location=a2m, use_weights=True, weights_location=wfile)   # computes + saves
    out = {"names": np.array(list(proc.seq_name_to_weight.keys())), "weights": np.asarray(proc.weights, np.float64),
           "Neff": np.float64(proc.Neff), "msa_start_end": np.array([ms - 1, me])}
    ck = os.path.join(HERE, "Tranception_toy")
    retr = dict(retrieval_aggregation_mode="aggregate_substitution", MSA_filename=a2m, full_protein_length=len(seq),
                MSA_weight_file_name=wfile, retrieval_inference_weight=0.6, MSA_start=ms - 1, MSA_end=me)
    model_r, tok = rh.reference_tranception_model(ck, retrieval=retr)
    out["msa_prior"] = msa_utils.get_msa_prior(MSA_data_file=a2m, MSA_weight_file_name=wfile, MSA_start=ms - 1, MSA_end=me,
                                               len_target_seq=len(seq), vocab=tok.get_vocab(),
                                               retrieval_aggregation_mode="aggregate_substitution")
    dms = pd.read_csv(os.path.join(HERE, "TOY_TRANCEPTION_DMS.csv"))
    r = model_r.score_mutants(DMS_data=dms, target_seq=seq, scoring_mirror=True, batch_size_inference=7, num_workers=0,
                              indel_mode=False)
    r = pd.merge(dms[["mutated_sequence"]], r, on="mutated_sequence", how="left")
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        out[f"scores_retrieval_weighted/{c}"] = r[c].to_numpy()
    np.savez_compressed(os.path.join(HERE, "golden_msa_weights.npz"), **out)
    print("kept", len(out["names"]), "of 91 sequences; Neff", out["Neff"])


if __name__ == "__main__":
    main()
