"""Golden fixtures for Tranception INDEL scoring WITH inference-time retrieval, produced by the UNMODIFIED reference
(tranception/model_pytorch.py:794-840 + tranception/utils/msa_utils.py:141-192) run on CPU through oracle/ref_harness.py.

The reference re-aligns every scored sequence to the family alignment with the user's Clustal Omega executable.  That binary is not in
this image; the reference is given tests/golden/stand_in_clustalo.py instead (a deterministic profile-to-sequence aligner that takes the same
command line), through a Bio.Align.Applications.ClustalOmegaCommandline stand-in that builds Biopython's command line (Biopython is not
installed either).  What is pinned is therefore everything AROUND the aligner: the files written for it, the walk over the two aligned rows,
the edited log-prior (a row dropped per deleted residue, a zero row per inserted one), MSA_end = MSA_start + rows, the fusion that leaves
inserted positions to the network alone, both directions, wild-type delta.

    python tests/golden/make_golden_tranception_indel_retrieval.py

Outputs: TOY_MSA_INDEL_FULL.a2m (covers the 70-residue target 1-70), TOY_TRANCEPTION_INDEL_RETRIEVAL_DMS.csv and
golden_tranception_indel_retrieval.npz:
  seq                                     the target
  full/<col>                              score_mutants(indel_mode=True) columns with the full-length alignment (MSA_start 0)
  partial_alignment                       what the reference does with TOY_MSA.a2m (residues 11-60, MSA_start 10): its row arithmetic assumes an
                                          alignment that covers the scored window, and it raises IndexError otherwise
  aligned/<k>                             the two aligned rows (sequence to score, reference) the stand-in returned for sequence k (full case)
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import ref_harness as rh  # noqa: E402

AA = "ACDEFGHIKLMNPQRSTVWY"
ALIGNER = os.path.join(HERE, "stand_in_clustalo.py")


def indel_library(rng, seq):
    """Deletions, insertions, both, a substitution, the wild type itself; distinct sequences."""
    out = [seq]
    L = len(seq)
    for _ in range(5):                                   # deletions of 1-3 residues
        p, n = int(rng.integers(1, L - 4)), int(rng.integers(1, 4))
        out.append(seq[:p] + seq[p + n:])
    for _ in range(5):                                   # insertions of 1-3 residues
        p = int(rng.integers(1, L - 1))
        out.append(seq[:p] + "".join(rng.choice(list(AA), size=int(rng.integers(1, 4)))) + seq[p:])
    for _ in range(3):                                   # a deletion and an insertion
        p, q = sorted(int(v) for v in rng.choice(np.arange(2, L - 3), size=2, replace=False))
        out.append(seq[:p] + seq[p + 1:q] + "".join(rng.choice(list(AA), size=2)) + seq[q:])
    out.append(seq[:1] + seq[2:])                        # near the ends
    out.append(seq[:-1] + "WW")
    s = list(seq)
    s[30] = "A" if s[30] != "A" else "C"
    out.append("".join(s))                               # a substitution scored in indel mode
    return list(dict.fromkeys(out))


def run(ck, a2m, ms, me, seq, dms):
    work = tempfile.mkdtemp()                            # the reference writes <MSA folder>/Sampled/*: keep the golden folder clean
    local = shutil.copy(a2m, os.path.join(work, os.path.basename(a2m)))
    retr = dict(retrieval_aggregation_mode="aggregate_indel", MSA_filename=local, full_protein_length=len(seq), MSA_weight_file_name=None,
                retrieval_inference_weight=0.6, MSA_start=ms, MSA_end=me, clustal_omega_location=ALIGNER)
    model, _ = rh.reference_tranception_model(ck, retrieval=retr)
    r = model.score_mutants(DMS_data=dms, target_seq=seq, scoring_mirror=True, batch_size_inference=1, num_workers=0, indel_mode=True)
    key = r["mutated_sequence"].fillna(r["mutant"]) if "mutant" in r else r["mutated_sequence"]       # the zero row sits under 'mutant' (:915-924)
    r = r.assign(key=key)
    m = pd.merge(dms[["mutated_sequence"]], r, left_on="mutated_sequence", right_on="key", how="left")
    return {c: m[c].to_numpy(dtype=np.float64) for c in ("avg_score_L_to_R", "avg_score_R_to_L", "avg_score")}, model, work


def main():
    rng = np.random.default_rng(4242)
    g = np.load(os.path.join(HERE, "golden_tranception.npz"))
    seq = str(g["seq"])
    ck = os.path.join(HERE, "Tranception_toy")
    lines = [">TARGET/1-70", seq]
    for i in range(30):
        s = list(seq)
        for p in rng.choice(len(s), size=int(rng.integers(4, 25)), replace=False):
            s[p] = rng.choice(list(AA + "-"))
        lines += [f">fam{i}/1-70", "".join(s)]
    full = os.path.join(HERE, "TOY_MSA_INDEL_FULL.a2m")
    open(full, "w").write("\n".join(lines) + "\n")
    lib = indel_library(rng, seq)
    dms = pd.DataFrame({"mutant": lib, "mutated_sequence": lib, "DMS_score": rng.standard_normal(len(lib))})
    dms["DMS_score_bin"] = (dms["DMS_score"] > 0).astype(int)
    dms.to_csv(os.path.join(HERE, "TOY_TRANCEPTION_INDEL_RETRIEVAL_DMS.csv"), index=False)
    out = {"seq": np.array(seq)}
    cols, model, work = run(ck, full, 0, len(seq), seq, dms)
    for c, v in cols.items():
        out[f"full/{c}"] = v
    from tranception.utils import msa_utils
    sampled = os.path.join(work, "Sampled")
    for k, s in enumerate(lib):                          # the aligner's answers, for tests that do not want to run it
        exp = os.path.join(sampled, f"probe_{k}.fa")
        open(os.path.join(sampled, "probe_seq.fa"), "w").write(">SEQ_TO_SCORE\n" + s + "\n")
        samp = [f for f in os.listdir(sampled) if f.startswith("Sampled_")][0]
        os.system(f"{ALIGNER} --profile1 {os.path.join(sampled, samp)} --profile2 {os.path.join(sampled, 'probe_seq.fa')} -o {exp} --force")
        d = msa_utils.process_msa_data(exp)
        out[f"aligned/{k}"] = np.array([d[">SEQ_TO_SCORE"], d[">REFERENCE_SEQUENCE"]])
    shutil.rmtree(work)
    try:                                                 # an alignment that does not start at residue 1 (TOY_MSA.a2m: 11-60)
        run(ck, os.path.join(HERE, "TOY_MSA.a2m"), 10, 60, seq, dms)
        out["partial_alignment"] = np.array("scored")
    except IndexError as e:                              # the mask over the prior slice (+1) and the scored positions differ in length
        out["partial_alignment"] = np.array("IndexError: " + str(e))
    np.savez_compressed(os.path.join(HERE, "golden_tranception_indel_retrieval.npz"), **out)
    print("full", np.round(out["full/avg_score"], 4))
    print("partial alignment:", out["partial_alignment"])


if __name__ == "__main__":
    main()
