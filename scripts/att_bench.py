"""Attention-kernel timing: ms per launch and algorithmic TFLOP/s of the attention class at several (sequences, tokens) shapes of the
ESM-1v 650M layer (median over rounds, measured by the library's own per-class HIP events).  Two builds of the kernel are compared with
scripts/lib_ab.sh; launch options of ONE build (pgmi_set_option) are compared here, interleaved round by round.

    python scripts/att_bench.py [--rounds 5] [--ab att_xcd_local=0,att_xcd_local=1]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteingym_amd import _lib, esm as pesm, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--shapes", default="286x286,90x1100,150x150,600x120")      # positions x residues
    ap.add_argument("--ab", default="", help="comma-separated option=value settings, timed alternately in every round (pgmi_set_option)")
    a = ap.parse_args()
    cfg = dict(synthetic.ESM1V_650M, layers=a.layers)
    model = pesm.EsmModel(cfg, synthetic.random_weights(cfg, seed=1), device=0)
    out = {}
    for shp in a.shapes.split(","):
        P, L = (int(v) for v in shp.split("x"))
        rng = np.random.default_rng(L)
        seq = synthetic.random_sequence(rng, L)
        pos = np.sort(rng.choice(L, size=min(P, L), replace=False))
        muts = [f"{seq[p]}{p + 1}{'A' if seq[p] != 'A' else 'C'}" for p in pos]
        if P > L:                                   # more sequences than residues: several assays' worth is emulated by --all-positions-like repeats
            muts = muts * (P // L + 1)
        assay = pesm.Assay(model, seq, muts)
        for v in [v for v in a.ab.split(",") if v]:              # the warm-up run on the defaults
            _lib.check(_lib.load().pgmi_set_option(v.split("=")[0].encode(), -1))
        assay.run_device_only()
        settings = [v for v in a.ab.split(",") if v] or [""]
        res = {v: [] for v in settings}
        for _ in range(a.rounds):
            for v in settings:
                if v:
                    name, val = v.split("=")
                    _lib.check(_lib.load().pgmi_set_option(name.encode(), int(val)))
                model.profile_reset()
                model.profile_enable(True)
                try:
                    assay.run_device_only()
                except _lib.PgmiError as e:                 # the timing probes of the two-role kernel (att_pp > 1) compute garbage: the range
                    if e.code != _lib.EOVERFLOW:            # guard at the end of the forward trips, the launches have run and been timed
                        raise
                model.profile_enable(False)
                pr = model.profile()["attention"]
                res[v].append((pr["ms"] / pr["launches"], pr["flops"] / (pr["ms"] * 1e-3) / 1e12))
        for v in settings:
            ms, tf = float(np.median([r[0] for r in res[v]])), float(np.median([r[1] for r in res[v]]))
            out[shp + (" " + v if v else "")] = {"T": assay.T, "sequences": len(assay.positions), "ms_per_launch": round(ms, 4), "tflops": round(tf, 1)}
            print(f"{shp:>10s} T={assay.T:4d} seqs={len(assay.positions):4d} {v:>18s}: {ms:.4f} ms/launch  {tf:6.1f} TFLOP/s", flush=True)
        assay.close()
    print(json.dumps(out))
    model.close()


if __name__ == "__main__":
    main()
