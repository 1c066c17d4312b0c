"""Secondary measurement (not the headline metric): Tranception-L-shaped scoring throughput.
BLAT_ECOLX-shaped assay (L=286), `--mutants` single mutants, both directions (scoring mirror),
synthetic weights.  Prints mutants/s and the per-kernel HIP-event breakdown."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import pandas as pd

sys.path.insert(0, os.getcwd())
from proteingym_amd import _lib, synthetic, tranception as ptr

ap = argparse.ArgumentParser()
ap.add_argument("--mutants", type=int, default=512)
ap.add_argument("--layers", type=int, default=36)
ap.add_argument("--doubles", action="store_true", help="a pairwise double-mutant library (24 x 3 first substitutions, each with 12 x 2 second ones = 1 728 "
                                                       "doubles) instead of single mutants: every sequence in full / the wild type as the root / intermediate roots")
args = ap.parse_args()
cfg = dict(synthetic.TRANCEPTION_L, layers=args.layers)
blob = synthetic.random_tranception_weights(cfg, seed=3)
model = ptr.TranceptionModel(cfg, blob, device=0)
seq, muts, _ = synthetic.random_assay(seed=23, L=286, n_single=args.mutants, n_multi=0)
if args.doubles:
    import numpy as np
    rng = np.random.default_rng(5)
    first = sorted(int(p) for p in rng.choice(len(seq) // 2, size=24, replace=False))
    second = sorted(int(p) for p in len(seq) // 2 + rng.choice(len(seq) // 2, size=12, replace=False))
    other = lambda p, k: [c for c in ptr.AA_vocab if c != seq[p]][k]   # noqa: E731
    muts = [f"{seq[i]}{i + 1}{other(i, a)}:{seq[j]}{j + 1}{other(j, b)}" for i in first for a in range(3) for j in second for b in range(2)]
df = pd.DataFrame({"mutant": muts})
df["mutated_sequence"] = ptr.mutated_sequences(seq, df["mutant"])
df = df.drop_duplicates("mutated_sequence")
ap_share = os.environ.get("PGMI_TR_SHARE_PREFIX", "1") != "0"
model.score_mutants(DMS_data=df.iloc[:32], target_seq=seq)          # warm-up
lib = _lib.load()
lines = {}
modes = ((False, True) if ap_share else (False,)) if not args.doubles else (False, "wild type root only", True)
for share in modes:             # every sequence in full (the reference's loop), then prefix-shared
    model.share_prefix = bool(share)
    model.share_intermediate = share is True
    model.rows_forwarded = model.rows_full = 0
    _lib.check(lib.pgmi_profile_reset(model._h)); _lib.check(lib.pgmi_profile_enable(model._h, 1))
    t0 = time.perf_counter()
    out = model.score_mutants(DMS_data=df, target_seq=seq, scoring_mirror=True)
    dt = time.perf_counter() - t0
    _lib.check(lib.pgmi_profile_enable(model._h, 0))
    prof = {}
    for k, name in enumerate(_lib.K_NAMES):
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(lib.pgmi_profile_get(model._h, k, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        prof[name] = dict(ms=round(ms.value, 2), tflops=round(fl.value / ms.value / 1e9, 1) if ms.value > 0 and fl.value > 0 else None)
    lines[share] = dict(mutants_per_s=len(df) / dt, seconds=dt, rows_forwarded=model.rows_forwarded, rows_of_the_full_forwards=model.rows_full,
                        tokens_per_s=model.rows_forwarded / dt, kernels=prof, scores=out["avg_score"].to_numpy())
same = bool(ap_share and all((v["scores"] == lines[False]["scores"]).all() for v in lines.values()))
for v in lines.values():
    del v["scores"]
best = lines[True] if ap_share else lines[False]
print(json.dumps({"metric": "mutants scored/sec (Tranception-L shape, no retrieval, both directions)", "value": best["mutants_per_s"],
                  "library": "pairwise double mutants" if args.doubles else "single mutants",
                  "mutants": len(df), "layers": args.layers, "prefix_shared": lines.get(True), "every_sequence_in_full": lines[False],
                  **({"wild_type_root_only": lines["wild type root only"]} if args.doubles else {}), "same_bits": same}))
