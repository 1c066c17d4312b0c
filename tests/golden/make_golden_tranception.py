"""Golden fixtures for the Tranception path, produced by the UNMODIFIED reference
(/root/reference/proteingym/baselines/tranception) run on CPU through the import shims of
oracle/ref_harness.py (transformers 5.x compatibility only; no arithmetic is touched).

    python tests/golden/make_golden_tranception.py

Outputs: Tranception_toy/ (config.json + pytorch_model.bin: 2 layers, n_embd 256, 4 heads = head_dim
64, built by the reference constructor), TOY_TRANCEPTION_DMS.csv, TOY_TRANCEPTION_LONG_DMS.csv,
TOY_MSA.a2m, golden_tranception.npz:
  logits / logits_ids / logits_mask          model(**batch).logits for a padded batch
  scores/<col>, scores_long/<col>             score_mutants() (mirror, optimal window), no retrieval
  scores_retrieval/<col>                      same with inference-time retrieval (alpha 0.6, MSA prior)
  msa_prior                                   msa_utils.get_msa_prior output
"""
import os
import sys

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import ref_harness as rh  # noqa: E402

AA = "ACDEFGHIKLMNPQRSTVWY"


def synth_dms(rng, seq, n_single, n_multi, su, include_wt=False):
    rows = []
    L = len(seq)
    for _ in range(n_single):
        p = int(rng.integers(0, L))
        rows.append(f"{seq[p]}{p + 1}{rng.choice([a for a in AA if a != seq[p]])}")
    for _ in range(n_multi):
        ps = sorted(rng.choice(L, size=int(rng.integers(2, 5)), replace=False))
        rows.append(":".join(f"{seq[p]}{p + 1}{rng.choice([a for a in AA if a != seq[p]])}" for p in ps))
    rows = list(dict.fromkeys(rows))
    df = pd.DataFrame({"mutant": rows})
    df["mutated_sequence"] = df["mutant"].apply(lambda x: su.get_mutated_sequence(seq, x))
    df["DMS_score"] = rng.standard_normal(len(df))
    df["DMS_score_bin"] = (df["DMS_score"] > 0).astype(int)
    return df


def main():
    rng = np.random.default_rng(777)
    ck = rh.make_tranception_checkpoint(os.path.join(HERE, "Tranception_toy"), 2, 256, 4, seed=1)
    sys.path.insert(0, rh.REF_TRANCEPTION)
    from tranception.utils import scoring_utils as su, msa_utils
    seq = "".join(rng.choice(list(AA), size=70))
    seq_long = "".join(rng.choice(list(AA), size=1100))
    dms = synth_dms(rng, seq, 25, 15, su)
    dms.to_csv(os.path.join(HERE, "TOY_TRANCEPTION_DMS.csv"), index=False)
    dms_long = synth_dms(rng, seq_long, 10, 6, su)
    dms_long.to_csv(os.path.join(HERE, "TOY_TRANCEPTION_LONG_DMS.csv"), index=False)
    out = {"seq": np.array(seq), "seq_long": np.array(seq_long)}

    model, tok = rh.reference_tranception_model(ck)
    batch = tok([seq, seq[:33], seq[::-1][:50]], add_special_tokens=True, truncation=True, padding=True, max_length=1024,
                return_tensors="pt")
    with torch.no_grad():
        lg = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], return_dict=True).logits
    out["logits"] = lg.numpy()
    out["logits_ids"] = batch["input_ids"].numpy()
    out["logits_mask"] = batch["attention_mask"].numpy()

    def run(model, df, target):
        r = model.score_mutants(DMS_data=df, target_seq=target, scoring_mirror=True, batch_size_inference=7,
                                num_workers=0, indel_mode=False)
        r = pd.merge(df[["mutated_sequence"]], r, on="mutated_sequence", how="left")
        return {c: r[c].to_numpy() for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score")}
    for c, v in run(model, dms, seq).items():
        out[f"scores/{c}"] = v
    for c, v in run(model, dms_long, seq_long).items():
        out[f"scores_long/{c}"] = v

    # retrieval: a small synthetic a2m covering residues 11..60 (1-indexed, inclusive) of the target
    ms, me = 11, 60
    focus = seq[ms - 1:me]
    lines = [">TARGET/11-60", focus]
    for i in range(40):
        s = list(focus)
        for p in rng.choice(len(s), size=int(rng.integers(3, 18)), replace=False):
            s[p] = rng.choice(list(AA + "-"))
        lines += [f">seq{i}/1-50", "".join(s)]
    lines += [">far/1-50", "".join(rng.choice(list(AA), size=len(focus)))]       # filtered out (<0.2 similarity, usually)
    a2m = os.path.join(HERE, "TOY_MSA.a2m")
    open(a2m, "w").write("\n".join(lines) + "\n")
    retr = dict(retrieval_aggregation_mode="aggregate_substitution", MSA_filename=a2m, full_protein_length=len(seq),
                MSA_weight_file_name=None, retrieval_inference_weight=0.6, MSA_start=ms - 1, MSA_end=me)
    model_r, _ = rh.reference_tranception_model(ck, retrieval=retr)
    out["msa_prior"] = msa_utils.get_msa_prior(MSA_data_file=a2m, MSA_weight_file_name=None, MSA_start=ms - 1, MSA_end=me,
                                               len_target_seq=len(seq), vocab=tok.get_vocab(),
                                               retrieval_aggregation_mode="aggregate_substitution")
    out["msa_start_end"] = np.array([ms - 1, me])
    for c, v in run(model_r, dms, seq).items():
        out[f"scores_retrieval/{c}"] = v
    np.savez_compressed(os.path.join(HERE, "golden_tranception.npz"), **out)
    print("wrote golden_tranception.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
