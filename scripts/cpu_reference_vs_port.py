#!/usr/bin/env python
"""Build-container only (needs /root/reference): times the UNMODIFIED reference model (fair-esm ProteinBertModel through
oracle/ref_harness.py) and the CPU port that bench.py's cpu_baseline uses (oracle/esm_oracle.py) side by side, same
ESM-1v-650M-shaped random checkpoint, same masked BLAT-length input, same thread count -- and checks that they agree.
bench.py's `cpu_baseline.kind` is "port" because the reference tree does not exist on the GPU box; this file is the
evidence that the port's speed is the reference's.

    python scripts/cpu_reference_vs_port.py > profiles/r2/cpu_reference_vs_port.json
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from oracle import esm_oracle as eo, ref_harness as rh  # noqa: E402
from proteingym_amd import synthetic  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count())
    cfg = dict(synthetic.ESM1V_650M)
    blob = synthetic.random_weights(cfg, seed=1, embed_std=0.15)
    seq, _, _ = synthetic.random_assay(seed=23, L=286, n_single=10, n_multi=0)
    with tempfile.TemporaryDirectory() as d:
        path = synthetic.save_fair_esm_checkpoint(os.path.join(d, "esm1v_synth_1.pt"), cfg, blob)
        model, alphabet = rh.reference_model(path)
        ocfg, W = eo.load_checkpoint(path)
    _, _, toks = alphabet.get_batch_converter()([("p", seq)])
    rows = {}

    def ref_fwd(i):
        t = toks.clone()
        t[0, i] = alphabet.mask_idx
        t0 = time.perf_counter()
        with torch.no_grad():
            lp = torch.log_softmax(model(t)["logits"], dim=-1)[0, i]
        return time.perf_counter() - t0, lp.numpy()

    def port_fwd(i):
        t = eo.tokenize(seq).copy()
        t[i] = eo.MASK
        t0 = time.perf_counter()
        with torch.no_grad():
            lp = torch.log_softmax(eo.forward_logits(ocfg, W, t[None]), dim=-1)[0, i]
        return time.perf_counter() - t0, lp.numpy()

    ref_fwd(1); port_fwd(1)                                  # warm-up
    tr, tp, err = [], [], 0.0
    for i in (5, 77, 150, 222, 280):
        a, ra = ref_fwd(i)
        b, rb = port_fwd(i)
        tr.append(a); tp.append(b)
        err = max(err, float(np.abs(ra - rb).max()))
    n_tok = toks.shape[1]
    print(json.dumps({"what": "batch-1 masked forward, ESM-1v 650M shape, T=288, CPU fp32",
                      "threads": torch.get_num_threads(), "host": "build container (no GPU)",
                      "reference_s_per_forward": float(np.median(tr)), "port_s_per_forward": float(np.median(tp)),
                      "reference_mutants_per_s_blat": 4996 / (float(np.median(tr)) * n_tok),
                      "port_mutants_per_s_blat": 4996 / (float(np.median(tp)) * n_tok),
                      "max_abs_logprob_difference": err,
                      "note": "the reference runs all L+2 = 288 positions one forward each (compute_fitness.py:489-503)"}, indent=1))


if __name__ == "__main__":
    main()
