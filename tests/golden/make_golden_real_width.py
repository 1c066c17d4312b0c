"""Golden fixtures at BASELINE config 5's OWN shape: ESM2-650M pseudo-perplexity of ~735-residue members.

A 733-term pseudo-ppl sum at the 650M width is 1 466 batch-1 forwards of ~737 tokens for two members: ~40 CPU-minutes,
far too slow for the test run, so the UNMODIFIED reference model (/root/reference/proteingym/baselines/esm/esm/model/
esm2.py:76-143, loaded by esm/pretrained.py:24-28) is driven here once, with the reference's own loop
(compute_fitness.py:258-279: token i masked, sequence[i] looked up, range(1, len-1), batch 1), and the per-position
terms are frozen.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_real_width.py [--fp64-members 1]

Weights are NOT stored: they are ``synthetic.random_weights(ESM2_650M, seed=5, embed_std=0.15)`` (numpy PCG64, the same
bits on every box of this image); the script writes them as a fair-esm v2 file into a temporary folder for the
reference loader.  Sequences: ``synthetic.random_indel_library(7, 735, 2)`` (the members bench.py's config-5 leg uses).

Output (committed): tests/golden/golden_pppl_650m.npz
    seq/<r>                the member
    terms/<r>              float64 array of the reference model's fp32 terms, i = 1 .. len-2
    sum/<r>                python-float left-to-right sum of the terms (what compute_pppl returns)
    terms64/<r>            (first --fp64-members members) the same terms from oracle/esm_oracle.py in float64: the
                           reference arithmetic's own rounding noise on a 733-term sum = |sum(terms) - sum(terms64)|
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402
from oracle import esm_oracle as eo  # noqa: E402
from proteingym_amd import synthetic  # noqa: E402

SEED, EMBED_STD, LIB_SEED, L = 5, 0.15, 7, 735


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fp64-members", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(HERE, "golden_pppl_650m.npz"))
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = dict(synthetic.ESM2_650M)
    blob = synthetic.random_weights(cfg, seed=SEED, embed_std=EMBED_STD)
    seqs = synthetic.random_indel_library(LIB_SEED, L, 2)[1]
    out = {}
    with tempfile.TemporaryDirectory() as d:
        path = synthetic.save_fair_esm_checkpoint(os.path.join(d, "esm2_t33_650M_synth.pt"), cfg, blob)
        model, alphabet = rh.reference_model(path)
    conv = alphabet.get_batch_converter()
    for r, s in enumerate(seqs):
        _, _, toks = conv([("protein1", s)])
        terms = []
        t0 = time.time()
        for i in range(1, len(s) - 1):                                  # compute_fitness.py:262-277
            t = toks.clone()
            t[0, i] = alphabet.mask_idx
            with torch.no_grad():
                lp = torch.log_softmax(model(t)["logits"], dim=-1)
            terms.append(lp[0, i, alphabet.get_idx(s[i])].item())
            if i % 50 == 0:
                print(f"member {r}: {i}/{len(s) - 2} terms, {time.time() - t0:.0f}s", flush=True)
        out[f"seq/{r}"] = np.array(s)
        out[f"terms/{r}"] = np.array(terms, dtype=np.float64)
        out[f"sum/{r}"] = np.array(sum(terms))
        np.savez_compressed(args.out, **out)
    del model
    ocfg, W = eo.from_arrays(arrays=synthetic.blob_to_arrays(cfg, blob), dtype=torch.float64, **cfg)
    for r, s in enumerate(seqs[: args.fp64_members]):
        tokens = eo.tokenize(s)[None, :]
        terms = []
        t0 = time.time()
        B = 2
        idx = list(range(1, len(s) - 1))
        with torch.no_grad():
            for c in range(0, len(idx), B):
                chunk = idx[c:c + B]
                t = np.repeat(tokens, len(chunk), axis=0)
                for b, i in enumerate(chunk):
                    t[b, i] = eo.MASK
                lp = torch.log_softmax(eo.forward_logits(ocfg, W, t), dim=-1)
                for b, i in enumerate(chunk):
                    terms.append(lp[b, i, eo.get_idx(s[i])].item())
                if c % 50 == 0:
                    print(f"fp64 member {r}: {c}/{len(idx)} terms, {time.time() - t0:.0f}s", flush=True)
        out[f"terms64/{r}"] = np.array(terms, dtype=np.float64)
        noise = abs(float(out[f"sum/{r}"]) - sum(terms))
        print(f"member {r}: reference fp32 sum vs fp64 sum differ by {noise:.3e}; per-term max "
              f"{np.abs(out[f'terms/{r}'] - out[f'terms64/{r}']).max():.3e}")
        np.savez_compressed(args.out, **out)
    print("wrote", args.out, sorted(out))


if __name__ == "__main__":
    main()
