#!/usr/bin/env python
"""BASELINE config 5 on one GPU: pseudo-perplexity of a DMS_indels.csv-shaped synthetic slice, ESM2-650M shape.

    python scripts/bench_pppl.py [--capsd 4] [--small-assays 12] [--precision f16x3]

Slice = `--capsd` members of a CAPSD_AAV2S-shaped library (735 residues; 733 masked forwards of ~737 tokens each: the
assay that holds 79 % of the benchmark's mutants and 99 % of its FLOPs) + every mutant of the `--small-assays` smallest
assays (37-72 residues).  Reports mutants/s, masked forwards/s, tokens/s, algorithmic TFLOP/s and packing efficiency
(real / padded tokens of the length-mixed batches).  The full benchmark (287 207 mutants, 1.95e8 forwards) is
extrapolated from the two rates; it is ~2e5 PFLOP of algorithmic work."""
import argparse
import json
import os
import sys
import time


sys.path.insert(0, os.getcwd())
from proteingym_amd import dist as pdist, esm as pesm, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--capsd", type=int, default=4)
    ap.add_argument("--small-assays", type=int, default=12)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--layers", type=int, default=33)
    args = ap.parse_args()
    cfg = dict(synthetic.ESM2_650M, layers=args.layers)
    model = pesm.EsmModel(cfg, synthetic.random_weights(cfg, seed=1, embed_std=0.15), device=0, precision=args.precision)
    shapes = sorted(synthetic.indel_shapes(), key=lambda r: r["n_total"])
    out = {"metric": "mutants scored/sec (ESM2-650M pseudo-ppl, indel libraries)", "precision": args.precision, "parts": {}}
    total_fl = lambda seqs: sum(max(0, len(s) - 2) * pdist.forward_flops(len(s) + 2, layers=args.layers) for s in seqs)
    for name, seqs in (("capsd_shaped", synthetic.random_indel_library(7, 735, args.capsd)[1]),
                       ("small_assays", [s for k, r in enumerate(shapes[:args.small_assays])
                                         for s in synthetic.random_indel_library(100 + k, r["seq_len"], r["n_total"])[1]])):
        lib = pesm.SequenceLibrary(model, seqs)
        lib.score(first=0, count=1)                                # warm-up (rotary tables, first-touch)
        t0 = time.perf_counter()
        lib.score()
        dt = time.perf_counter() - t0
        st = lib.stats()
        out["parts"][name] = {"sequences": len(seqs), "lengths": [min(map(len, seqs)), max(map(len, seqs))], "seconds": round(dt, 3),
                              "mutants_per_s": len(seqs) / dt, "forwards_per_s": st["rows"] / dt, "tokens_per_s": st["tokens"] / dt,
                              "algorithmic_tflops": total_fl(seqs) / dt / 1e12, "batches": st["batches"],
                              "packing_efficiency": st["packing_efficiency"]}
        lib.close()
    big = [r for r in synthetic.indel_shapes()]
    est = 0.0
    for r in big:                                                # whole benchmark at the measured token rates
        part = out["parts"]["capsd_shaped"] if r["seq_len"] > 150 else out["parts"]["small_assays"]
        est += r["n_total"] * max(0, r["seq_len"] - 2) * (r["seq_len"] + 2) / part["tokens_per_s"]
    out["full_benchmark_estimate"] = {"mutants": sum(r["n_total"] for r in big), "gpu_hours_1gpu": est / 3600,
                                      "gpu_hours_8gpu": est / 3600 / 8,
                                      "note": "token-rate extrapolation; attention share grows with length (rates taken per length class)"}
    print(json.dumps(out))
    model.close()


if __name__ == "__main__":
    main()
