"""MSA Transformer masked-marginals throughput at the esm_msa1b_t12_100M_UR50S shape on one MI355X.

    python scripts/bench_msa_transformer.py [--rows 400] [--cols 287] [--positions 8]

One "forward" = the whole sampled alignment (rows x cols tokens) with one masked cell, as
compute_fitness.py:380-394 runs it once per column.  Synthetic weights and tokens.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteingym_amd import msa_transformer as pmsa, synthetic, _lib  # noqa: E402
import ctypes as C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=400)
    ap.add_argument("--cols", type=int, default=287)
    ap.add_argument("--positions", type=int, default=8)
    ap.add_argument("--layers", type=int, default=12)
    a = ap.parse_args()
    cfg = dict(synthetic.MSA_1B, layers=a.layers)
    arrays = synthetic.random_msa_transformer_arrays(cfg, seed=1)
    rp, cp = (a.rows + 31) // 32 * 32, (a.cols + 31) // 32 * 32
    m = pmsa.MsaTransformerModel(cfg, pmsa.pack_state_dict(cfg, arrays), max_rows=rp * cp)
    rng = np.random.default_rng(0)
    tok = rng.integers(4, 30, size=(a.rows, a.cols)).astype(np.int64)
    tok[:, 0] = 0
    pos = np.linspace(1, a.cols - 1, a.positions).astype(int)
    m.masked_logprobs(tok, pos[:1], seq_len=a.cols - 1)                 # warm-up
    lib = _lib.load()
    lib.pgmi_profile_enable(m._h, 1)
    lib.pgmi_profile_reset(m._h)
    t0 = time.perf_counter()
    m.masked_logprobs(tok, pos, seq_len=a.cols - 1)
    dt = time.perf_counter() - t0
    D, F, L = cfg["embed_dim"], cfg["ffn_dim"], cfg["layers"]
    M = a.rows * a.cols
    flops = L * (2.0 * M * D * (2 * 4 * D + 2 * F) + 4.0 * a.cols * a.cols * a.rows * D + 4.0 * M * a.rows * D)
    prof = {}
    for k, name in enumerate(_lib.K_NAMES):
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(lib.pgmi_profile_get(m._h, k, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        if n.value:
            prof[name] = {"ms_per_forward": round(ms.value / a.positions, 3), "launches": n.value // a.positions,
                          "tflops": round(fl.value / (ms.value * 1e-3) / 1e12, 1) if fl.value else None}
    print(json.dumps({"metric": "MSA Transformer masked forwards/s (esm_msa1b shape)", "rows": a.rows, "cols": a.cols,
                      "layers": L, "forwards": int(a.positions), "ms_per_forward": dt / a.positions * 1e3,
                      "tflops_algorithmic": flops * a.positions / dt / 1e12,
                      "blat_like_assay_seconds_5_seeds": dt / a.positions * (a.cols - 1) * 5, "profile": prof}))
    m.close()


if __name__ == "__main__":
    main()
