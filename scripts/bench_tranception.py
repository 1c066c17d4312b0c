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
args = ap.parse_args()
cfg = dict(synthetic.TRANCEPTION_L, layers=args.layers)
blob = synthetic.random_tranception_weights(cfg, seed=3)
model = ptr.TranceptionModel(cfg, blob, device=0)
seq, muts, _ = synthetic.random_assay(seed=23, L=286, n_single=args.mutants, n_multi=0)
df = pd.DataFrame({"mutant": muts})
df["mutated_sequence"] = df["mutant"].apply(lambda m: ptr.get_mutated_sequence(seq, m))
df = df.drop_duplicates("mutated_sequence")
ap_share = os.environ.get("PGMI_TR_SHARE_PREFIX", "1") != "0"
model.score_mutants(DMS_data=df.iloc[:32], target_seq=seq)          # warm-up
lib = _lib.load()
lines = {}
for share in ((False, True) if ap_share else (False,)):             # every sequence in full (the reference's loop), then prefix-shared
    model.share_prefix = share
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
same = bool(ap_share and (lines[True]["scores"] == lines[False]["scores"]).all())
for v in lines.values():
    del v["scores"]
best = lines[True] if ap_share else lines[False]
print(json.dumps({"metric": "mutants scored/sec (Tranception-L shape, no retrieval, both directions)", "value": best["mutants_per_s"],
                  "mutants": len(df), "layers": args.layers, "prefix_shared": lines.get(True), "every_sequence_in_full": lines[False],
                  "same_bits": same}))
