"""The north-star metric on the full-size workload: the 217-assay-shaped ProteinGym substitution benchmark
(real sequence lengths 37..3423 and mutant counts, 2 465 767 mutants; synthetic sequences, DMS files and
ESM-1v-650M-shaped weights) scored end to end through the product runner (run_benchmark: checkpoint load, assay
upload, masked-marginals with optimal 1024 windows, RCCL gather when N > 1, CSVs written).

    python scripts/bench_217.py [--max-assays N] [--workdir DIR]            (1 GPU)
    python -m torch.distributed.run --nproc-per-node 8 ... scripts/bench_217.py     (one rank per GPU)

mutants/s = rows of all scored DMS files / wall time of run_benchmark.main (from before the checkpoint is read to
all CSVs written).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteingym_amd import synthetic, run_benchmark, dist as pdist  # noqa: E402

AA = np.array(list(synthetic.AA))


def make_assay(rng, L, n_single, n_multi):
    seq_idx = rng.integers(0, 20, size=L)
    seq = "".join(AA[seq_idx])

    def subs(pos):
        mt = (seq_idx[pos] + rng.integers(1, 20, size=pos.shape)) % 20          # uniform over the 19 non-WT letters
        return np.char.add(np.char.add(AA[seq_idx[pos]], (pos + 1).astype(str)), AA[mt])
    rows = list(subs(rng.integers(0, L, size=n_single)))
    if n_multi:
        depth = rng.integers(2, 6, size=n_multi)
        for d in depth:
            pos = np.sort(rng.choice(L, size=min(int(d), L), replace=False))
            rows.append(":".join(subs(pos)))
    score = rng.standard_normal(len(rows))
    return seq, pd.DataFrame({"mutant": rows, "DMS_score": score, "DMS_score_bin": (score > 0).astype(int)})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-assays", type=int, default=0, help="first N assays of the reference file (0 = all 217)")
    ap.add_argument("--workdir", default="/tmp/pgmi_bench217")
    ap.add_argument("--precision", default="f16x3")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    shapes = synthetic.dms_shapes()
    if a.max_assays:
        shapes = shapes[:a.max_assays]
    os.makedirs(os.path.join(a.workdir, "dms"), exist_ok=True)
    ckpt = os.path.join(a.workdir, "esm1v_synth_1.pt")
    mapping_csv = os.path.join(a.workdir, "mapping.csv")
    t_gen = time.time()
    if rank == 0:
        rows = []
        for s in shapes:
            rng = np.random.default_rng(int(s["DMS_index"].split("_")[-1]) if isinstance(s["DMS_index"], str) else s["DMS_index"])
            seq, df = make_assay(rng, s["seq_len"], s["n_single"], s["n_total"] - s["n_single"])
            df.to_csv(os.path.join(a.workdir, "dms", s["DMS_id"] + ".csv"), index=False)
            rows.append({"DMS_id": s["DMS_id"], "DMS_filename": s["DMS_id"] + ".csv", "target_seq": seq,
                         "DMS_total_number_mutants": len(df)})
        pd.DataFrame(rows).to_csv(mapping_csv, index=False)
        cfg = dict(synthetic.ESM1V_650M)
        synthetic.save_fair_esm_checkpoint(ckpt, cfg, synthetic.random_weights(cfg, seed=1))
        open(os.path.join(a.workdir, "ready"), "w").write("1")
    else:
        while not os.path.exists(os.path.join(a.workdir, "ready")):
            time.sleep(1)
    t_gen = time.time() - t_gen
    out_dir = os.path.join(a.workdir, "scores")
    args = run_benchmark.create_parser().parse_args([
        "--model-location", ckpt, "--model_type", "ESM1v", "--dms_mapping", mapping_csv,
        "--dms-input", os.path.join(a.workdir, "dms"), "--dms-output", out_dir, "--precision", a.precision,
        "--overwrite-prior-scores"])
    t0 = time.time()
    run_benchmark.main(args)
    dt = time.time() - t0
    if rank == 0:
        n_mut = sum(s["n_total"] for s in shapes)
        tokens = sum(min(s["seq_len"] + 2, 1024) * (s["seq_len"]) for s in shapes)
        flops = sum(pdist.assay_cost(s["seq_len"]) for s in shapes)
        some = pd.read_csv(os.path.join(out_dir, shapes[0]["DMS_id"] + ".csv"))
        print(json.dumps({"metric": "mutants scored/sec (ESM-1v 650M masked-marginal), 217-assay-shaped benchmark, 1 checkpoint",
                          "value": n_mut / dt, "unit": "mutants/s", "n_gpus": world, "assays": len(shapes), "mutants": n_mut,
                          "seconds": dt, "algorithmic_pflop_all_positions": flops / 1e15,
                          "tflops_if_all_positions_run": flops / dt / 1e12, "approx_tokens": tokens,
                          "setup_seconds_not_timed": t_gen, "precision": a.precision, "data": "synthetic",
                          "columns_of_first_csv": list(some.columns)}), flush=True)


if __name__ == "__main__":
    main()
