"""A/B of run_benchmark's short-assay groups on the short end of the 217-assay-shaped table (proteins of at most --tokens tokens
and at most --rows mutants; ESM-1v-650M-shaped synthetic checkpoint): the same assays scored one at a time and several at a time,
seconds inside the scorer and the achieved rate by protein length.  Prints one JSON object.

    python scripts/short_assay_ab.py [--tokens 200] [--rows 20000] [--rounds 2]
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from proteingym_amd import dist as pdist, run_benchmark, synthetic  # noqa: E402
import bench_217  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=200)
    ap.add_argument("--rows", type=int, default=20000)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    cfg = dict(synthetic.ESM1V_650M)
    shapes = [s for s in synthetic.dms_shapes() if s["seq_len"] + 2 <= a.tokens and s["n_total"] <= a.rows]
    out = {"assays": len(shapes), "mutants": int(sum(s["n_total"] for s in shapes)), "runs": []}
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "dms"))
        rows = []
        for sh in shapes:
            sq, df = bench_217.make_assay(np.random.default_rng(sh["DMS_index"]), sh["seq_len"], sh["n_single"], sh["n_total"] - sh["n_single"])
            df.to_csv(os.path.join(d, "dms", sh["DMS_id"] + ".csv"), index=False)
            rows.append({"DMS_id": sh["DMS_id"], "DMS_filename": sh["DMS_id"] + ".csv", "target_seq": sq, "DMS_total_number_mutants": len(df)})
        pd.DataFrame(rows).to_csv(os.path.join(d, "map.csv"), index=False)
        ck = synthetic.save_fair_esm_checkpoint(os.path.join(d, "esm1v_synth_1.pt"), cfg, synthetic.random_weights(cfg, seed=2))
        common = ["--model-location", ck, "--model_type", "ESM1v", "--dms_mapping", os.path.join(d, "map.csv"), "--dms-input", os.path.join(d, "dms")]
        for r in range(a.rounds):
            for tag, extra in (("one_at_a_time", ["--batch-short-tokens", "0"]), ("grouped", [])):
                o = os.path.join(d, f"out_{tag}_{r}")
                st = run_benchmark.main(run_benchmark.create_parser().parse_args(common + ["--dms-output", o, *extra]))
                log = st["rank0_assays"]
                fl = [e["positions_run"] * pdist.forward_flops(e["T"]) for e in log]
                hist = {}
                for lo, hi in ((0, 60), (60, 100), (100, 200)):
                    sel = [k for k, e in enumerate(log) if lo <= e["seq_len"] < hi]
                    if sel:
                        t = sum(log[k]["run_s"] for k in sel)
                        hist[f"{lo}-{hi - 1}"] = {"assays": len(sel), "run_s": round(t, 3), "executed_tflops": round(sum(fl[k] for k in sel) / max(t, 1e-9) / 1e12, 1)}
                out["runs"].append({"mode": tag, "round": r, "run_s": round(st["rank0_wall_clock"].get("assay_run_s", 0.0), 3),
                                    "create_s": round(st["rank0_wall_clock"].get("assay_create_s", 0.0), 3),
                                    "score_s": round(st["rank0_wall_clock"]["score_s"], 3), "wall_s": round(st["seconds"], 2),
                                    "groups": len({(e["padded_T"], e["group_of"]) for e in log if e.get("group_of", 1) > 1}),
                                    "by_residues": hist})
        same = all(open(os.path.join(d, "out_grouped_0", r["DMS_id"] + ".csv")).read() == open(os.path.join(d, "out_one_at_a_time_0", r["DMS_id"] + ".csv")).read()
                   for r in rows)
        out["csv_files_byte_identical"] = bool(same)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
