"""Golden fixture for the MSA Transformer's pseudo-ppl branch (compute_fitness.py:258-279 with mode == "MSA_Transformer",
driver :403-417): the UNMODIFIED reference CLI on the toy MSA checkpoint / alignment of make_golden_msa_transformer.py.

    python tests/golden/make_golden_msa_pppl.py          (build container only: needs /root/reference)

Per mutant the reference prepends the mutated sequence to the sampled alignment (which still holds the wild type as its first
row), masks token i of that new first row for i in range(1, len(sequence) - 1) and sums log p(sequence[i]) -- the same
off-by-one as the single-sequence compute_pppl.  Output (committed): golden_msa_pppl.npz
    cli/columns, cli/msa_toy_seed1, cli/msa_toy_seed2, cli/msa_toy_ensemble      the CSV the reference wrote
    mutants                                                                      the rows scored (single substitutions)
"""
import os
import sys
import tempfile

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import ref_harness as rh  # noqa: E402


def main():
    src = pd.read_csv(os.path.join(HERE, "TOY_MSA_DMS.csv"))
    singles = src[~src["mutant"].str.contains(":")].iloc[:3].reset_index(drop=True)
    out = {"mutants": np.array(list(singles["mutant"]))}
    with tempfile.TemporaryDirectory() as d:
        singles.to_csv(os.path.join(d, "TOY_MSA_PPPL.csv"), index=False)
        mp = pd.read_csv(os.path.join(HERE, "TOY_MSA_MAPPING.csv"))
        mp["DMS_id"], mp["DMS_filename"] = "TOY_MSA_PPPL", "TOY_MSA_PPPL.csv"
        mp.to_csv(os.path.join(d, "map.csv"), index=False)
        rh.run_reference_cli(["--model-location", os.path.join(HERE, "msa_toy.pt"), "--model_type", "MSA_transformer", "--dms_index", "0",
                              "--dms_mapping", os.path.join(d, "map.csv"), "--dms-input", d, "--dms-output", os.path.join(d, "o"),
                              "--scoring-strategy", "pseudo-ppl", "--msa-path", HERE, "--msa-weights-folder", HERE,
                              "--msa-samples", "64", "--seeds", "1", "2", "--nogpu"])
        df = pd.read_csv(os.path.join(d, "o", "TOY_MSA_PPPL.csv"))
    out["cli/columns"] = np.array(list(df.columns))
    for c in ("msa_toy_seed1", "msa_toy_seed2", "msa_toy_ensemble"):
        out[f"cli/{c}"] = df[c].to_numpy()
    np.savez_compressed(os.path.join(HERE, "golden_msa_pppl.npz"), **out)
    print(df)


if __name__ == "__main__":
    main()
