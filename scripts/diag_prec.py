import os, sys, numpy as np, pandas as pd
sys.path.insert(0, os.getcwd())
from proteingym_amd import esm as pesm
g = np.load("tests/golden/golden_esm.npz")
seq = str(g["seq"]); df = pd.read_csv("tests/golden/TOY_DMS.csv")
for name in ("esm1b_toy_lnb", "esm1v_toy_1", "esm2_toy"):
    for prec in ("fp32", "f16x3"):
        m = pesm.load_model_and_alphabet(f"tests/golden/{name}.pt", precision=prec)[0]
        s, t = pesm.Assay(m, seq, list(df["mutant"]), all_positions=True).run(want_table=True)
        ref = g[f"{name}/mm_table"]
        e = np.abs(t - ref)
        i = np.unravel_index(e.argmax(), e.shape)
        print(name, prec, "max", e.max(), "at", i, "ref", ref[i], "got", t[i], "p99", np.quantile(e, 0.99), "score err", np.abs(s - g[f"cli/{name}"]).max(),
              "AA cols max", e[:, 4:24].max())
        m.close()
