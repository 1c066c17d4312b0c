"""Causal (Tranception) attention class under a launch option, interleaved: ms per launch of the attention class (prep + kernel) for a
batch of full forwards at several protein lengths.     python scripts/att_bench_tranception.py [--ab att_xcd_local=0,att_xcd_local=1]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteingym_amd import _lib, synthetic, tranception as ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--lengths", default="150,286,380,500,900")
ap.add_argument("--ab", default="att_xcd_local=0,att_xcd_local=1")
a = ap.parse_args()
cfg = dict(synthetic.TRANCEPTION_L, layers=a.layers)
model = ptr.TranceptionModel(cfg, synthetic.random_tranception_weights(cfg, seed=3), device=0)
lib = _lib.load()
settings = [v for v in a.ab.split(",") if v]
for L in (int(v) for v in a.lengths.split(",")):
    rng = np.random.default_rng(L)
    n = max(8, 90000 // (L + 2))
    seqs = [synthetic.random_sequence(rng, L) for _ in range(n)]
    ids, _ = model.encode_batch(seqs)
    model.token_logprobs(ids)
    res = {v: [] for v in settings}
    for _ in range(a.rounds):
        for v in settings:
            name, val = v.split("=")
            _lib.check(lib.pgmi_set_option(name.encode(), int(val)))
            _lib.check(lib.pgmi_profile_reset(model._h)); _lib.check(lib.pgmi_profile_enable(model._h, 1))
            model.token_logprobs(ids)
            _lib.check(lib.pgmi_profile_enable(model._h, 0))
            import ctypes as C
            ms, cnt, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            _lib.check(lib.pgmi_profile_get(model._h, _lib.K_NAMES.index("attention"), C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by)))
            res[v].append(ms.value / max(cnt.value, 1))
    for v in settings:
        print(f"L={L:4d} T={ids.shape[1]:4d} seqs={n:4d} {v:>18s}: {float(np.median(res[v])):.4f} ms per attention scope (prep + kernel)", flush=True)
for v in settings:
    lib.pgmi_set_option(v.split("=")[0].encode(), -1)
model.close()
