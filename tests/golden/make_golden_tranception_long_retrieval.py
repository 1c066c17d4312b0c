"""Golden for inference-time retrieval on a protein LONGER than the context (1100 residues, optimal 1022-residue
windows) with an alignment that covers only residues 301..900: exercises the window/alignment index arithmetic of
the fusion in both directions (model_pytorch.py:806-830).  Produced by the REFERENCE on CPU.

    python tests/golden/make_golden_tranception_long_retrieval.py  -> TOY_MSA_LONGSPAN.a2m, golden_tranception_long_retrieval.npz
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
    rng = np.random.default_rng(4711)
    g = np.load(os.path.join(HERE, "golden_tranception.npz"))
    seq_long = str(g["seq_long"])
    ms, me = 301, 900                                     # 1-indexed inclusive
    focus = seq_long[ms - 1:me]
    lines = [f">LONGSPAN/{ms}-{me}", focus]
    for i in range(30):
        s = list(focus)
        for p in rng.choice(len(s), size=int(rng.integers(20, 200)), replace=False):
            s[p] = rng.choice(list(AA + "-"))
        lines += [f">h{i}/1-{len(focus)}", "".join(s)]
    a2m = os.path.join(HERE, "TOY_MSA_LONGSPAN.a2m")
    open(a2m, "w").write("\n".join(lines) + "\n")
    retr = dict(retrieval_aggregation_mode="aggregate_substitution", MSA_filename=a2m, full_protein_length=len(seq_long),
                MSA_weight_file_name=None, retrieval_inference_weight=0.6, MSA_start=ms - 1, MSA_end=me)
    model, _ = rh.reference_tranception_model(os.path.join(HERE, "Tranception_toy"), retrieval=retr)
    dms = pd.read_csv(os.path.join(HERE, "TOY_TRANCEPTION_LONG_DMS.csv"))
    r = model.score_mutants(DMS_data=dms, target_seq=seq_long, scoring_mirror=True, batch_size_inference=4, num_workers=0,
                            indel_mode=False)
    r = pd.merge(dms[["mutated_sequence"]], r, on="mutated_sequence", how="left")
    out = {"msa_start_end": np.array([ms - 1, me])}
    for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score"):
        out[f"scores/{c}"] = r[c].to_numpy()
    np.savez_compressed(os.path.join(HERE, "golden_tranception_long_retrieval.npz"), **out)
    print({k: v[:3] for k, v in out.items()})


if __name__ == "__main__":
    main()
