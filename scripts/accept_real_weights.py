#!/usr/bin/env python
"""Acceptance run on REAL checkpoints: the north star's second clause ("Spearman vs DMS identical ...") against the numbers ProteinGym
publishes.

    python scripts/accept_real_weights.py --proteingym /path/to/ProteinGym --dms-folder DMS_ProteinGym_substitutions \\
        --esm1v esm1v_t33_650M_UR90S_{1,2,3,4,5}.pt [--esm2 esm2_t33_650M_UR50D.pt esm2_t36_3B_UR50D.pt] \\
        [--tranception Tranception_Large [--msa-folder MSA_files --msa-weights-folder MSA_weights]] \\
        [--assays BLAT_ECOLX_Stiffler_2015] --out accept_out

What it does, with nothing of its own between the scorer and the verdict:
  1. cuts ``<proteingym>/reference_files/DMS_substitutions.csv`` down to ``--assays``;
  2. scores them with this package's runners (``run_benchmark`` for ESM-1v / ESM2, the single-assay Tranception CLI), writing the CSVs
     where ``<proteingym>/config.json`` says each model's scores live;
  3. runs the checkout's OWN ``proteingym/merge.py`` and ``proteingym/performance_DMS_benchmarks.py`` on them (the latter's summary
     tables need every taxon / alignment-depth class and may stop on a one-assay subset: its per-assay table is written first and is all
     that is read here; scipy's spearmanr on the merged file is printed beside it);
  4. compares the per-assay Spearman with ``<proteingym>/benchmarks/DMS_zero_shot/substitutions/Spearman/
     DMS_substitutions_Spearman_DMS_level.csv`` (3 decimals; BLAT_ECOLX_Stiffler_2015: ESM-1v single 0.668, ensemble 0.707, ESM2 650M
     0.731, ESM2 3B 0.589, Tranception L 0.629).
Exit code 0: every compared value within ``--tolerance`` (default 0.001 = the last published decimal); 1: a value differs; 2: an input
(checkpoint, DMS file, ProteinGym checkout) is absent -- this container and the GPU boxes hold neither weights nor DMS data.
``main(argv, make_model=..., make_tranception=...)`` are test seams (tests/test_oracle_pinning.py drives the whole flow on the toy goldens).
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PUBLISHED = os.path.join("benchmarks", "DMS_zero_shot", "substitutions", "Spearman", "DMS_substitutions_Spearman_DMS_level.csv")
FIELD = "model_list_zero_shot_substitutions_DMS"


def create_parser():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--proteingym", default=os.environ.get("PROTEINGYM_ROOT", ""), help="ProteinGym checkout (reference_files/, config.json, proteingym/merge.py ...)")
    ap.add_argument("--dms-folder", required=True, help="DMS_ProteinGym_substitutions (one CSV per assay)")
    ap.add_argument("--esm1v", nargs="*", default=[], help="ESM-1v checkpoints (.pt); the first is 'ESM-1v (single)', all of them the ensemble")
    ap.add_argument("--esm2", nargs="*", default=[], help="ESM2 checkpoints (.pt); matched to the registry by file stem")
    ap.add_argument("--tranception", default=None, help="Tranception checkpoint directory (config.json + weights)")
    ap.add_argument("--msa-folder", default=None, help="with --tranception: alignments -> 'Tranception L' (inference-time retrieval); without: 'Tranception L no retrieval'")
    ap.add_argument("--msa-weights-folder", default=None)
    ap.add_argument("--assays", nargs="+", default=["BLAT_ECOLX_Stiffler_2015"], help="DMS_id values")
    ap.add_argument("--out", default="accept_out")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--tolerance", type=float, default=0.001)
    ap.add_argument("--published", default=None, help="per-assay Spearman table to compare with (default: the checkout's)")
    return ap


def _load_script(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run_script_main(mod, argv):
    """mod.main() under sys.argv = argv; (ok, message)."""
    saved = sys.argv
    sys.argv = argv
    try:
        mod.main()
        return True, ""
    except SystemExit as e:
        return e.code in (None, 0), f"exit {e.code}"
    except Exception as e:                                      # noqa: BLE001 -- the caller decides what a failure of the reference's script means
        return False, f"{type(e).__name__}: {e}"
    finally:
        sys.argv = saved


def missing_inputs(args, reference, subset):
    """Every absent input, by name (exit code 2 lists them all at once)."""
    gone = []
    pg = args.proteingym
    if not pg or not os.path.isdir(pg):
        return [f"--proteingym {pg!r}: not a directory (a ProteinGym checkout is needed for reference_files/, config.json, merge.py, "
                "performance_DMS_benchmarks.py and the published Spearman table)"]
    for rel in ("reference_files/DMS_substitutions.csv", "config.json", "proteingym/merge.py", "proteingym/performance_DMS_benchmarks.py",
                "proteingym/constants.json"):
        if not os.path.exists(os.path.join(pg, rel)):
            gone.append(os.path.join(pg, rel))
    published = args.published or os.path.join(pg, PUBLISHED)
    if not os.path.exists(published):
        gone.append(published)
    if reference is not None:
        for a in args.assays:
            if a not in set(reference["DMS_id"]):
                gone.append(f"assay {a}: not a DMS_id of reference_files/DMS_substitutions.csv")
    if subset is not None:
        for fn in subset["DMS_filename"]:
            if not os.path.exists(os.path.join(args.dms_folder, str(fn))):
                gone.append(os.path.join(args.dms_folder, str(fn)))
    for p in list(args.esm1v) + list(args.esm2):
        if not os.path.exists(p):
            gone.append(p)
    if args.tranception and not os.path.exists(os.path.join(args.tranception, "config.json")):
        gone.append(os.path.join(args.tranception, "config.json"))
    if args.tranception and args.msa_folder and subset is not None and "MSA_filename" in subset:
        for fn in subset["MSA_filename"]:
            if not os.path.exists(os.path.join(args.msa_folder, str(fn))):
                gone.append(os.path.join(args.msa_folder, str(fn)))
    if not (args.esm1v or args.esm2 or args.tranception):
        gone.append("no model given: --esm1v / --esm2 / --tranception")
    return gone


def main(argv=None, make_model=None, make_tranception=None, patch_performance=None) -> int:
    args = create_parser().parse_args(argv)
    pg = args.proteingym
    ref_path = os.path.join(pg, "reference_files", "DMS_substitutions.csv") if pg else ""
    reference = pd.read_csv(ref_path) if pg and os.path.exists(ref_path) else None
    subset = reference[reference["DMS_id"].isin(args.assays)].reset_index(drop=True) if reference is not None else None
    gone = missing_inputs(args, reference, subset)
    if gone:
        print("accept_real_weights: cannot run, these inputs are absent:", file=sys.stderr)
        for g in gone:
            print("   " + g, file=sys.stderr)
        print("(real checkpoints and the DMS files are not part of this repository; nothing was scored)", file=sys.stderr)
        return 2
    os.makedirs(args.out, exist_ok=True)
    sub_csv = os.path.join(args.out, "DMS_substitutions_subset.csv")
    subset.to_csv(sub_csv, index=False)
    registry = json.load(open(os.path.join(pg, "config.json")))[FIELD]
    clean = json.load(open(os.path.join(pg, "proteingym", "constants.json")))["clean_names"]
    scores = os.path.join(args.out, "scores")
    models = {}                                                 # registry key -> registry entry (score column adjusted to the files given)

    from proteingym_amd import run_benchmark as rb
    os.environ.setdefault("LOCAL_RANK", str(args.device))      # run_benchmark takes its device from the launcher's environment
    stem = lambda p: os.path.splitext(os.path.basename(p))[0]    # noqa: E731
    if args.esm1v:
        models["ESM1v_single"] = dict(registry["ESM1v_single"], input_score_name=stem(args.esm1v[0]))
        if len(args.esm1v) > 1:
            models["ESM1v_ensemble"] = dict(registry["ESM1v_ensemble"])
        rb.main(rb.create_parser().parse_args(["--model-location", *args.esm1v, "--model_type", "ESM1v", "--dms_mapping", sub_csv, "--dms-input", args.dms_folder,
                                               "--dms-output", os.path.join(scores, registry["ESM1v_ensemble"]["location"])]),
                **({"make_model": make_model} if make_model else {}))
    for ckpt in args.esm2:
        key = next((k for k, v in registry.items() if k.startswith("ESM2") and v["input_score_name"] == stem(ckpt)), None)
        if key is None:
            print(f"accept_real_weights: {ckpt}: no ESM2 entry of config.json has the score column {stem(ckpt)!r}", file=sys.stderr)
            return 2
        models[key] = dict(registry[key])
        rb.main(rb.create_parser().parse_args(["--model-location", ckpt, "--model_type", "ESM2", "--dms_mapping", sub_csv, "--dms-input", args.dms_folder,
                                               "--dms-output", os.path.join(scores, registry[key]["location"])]),
                **({"make_model": make_model} if make_model else {}))
    if args.tranception:
        from proteingym_amd import score_tranception_proteingym as cli
        key = "Tranception_L" if args.msa_folder else "Tranception_L_no_retrieval"
        models[key] = dict(registry[key])
        extra = ["--inference_time_retrieval", "--MSA_folder", args.msa_folder] + \
                (["--MSA_weights_folder", args.msa_weights_folder] if args.msa_weights_folder else []) if args.msa_folder else []
        for i in range(len(subset)):
            targv = ["--checkpoint", args.tranception, "--DMS_reference_file_path", sub_csv, "--DMS_data_folder", args.dms_folder, "--DMS_index", str(i),
                     "--output_scores_folder", os.path.join(scores, registry[key]["location"]), "--device", str(args.device), *extra]
            if make_tranception:
                make_tranception(cli, cli.create_parser().parse_args(targv))
            else:
                cli.main(cli.create_parser().parse_args(targv))
    cfg_path = os.path.join(args.out, "config.json")
    json.dump({FIELD: models}, open(cfg_path, "w"), indent=1)

    merge = _load_script("pg_accept_merge", os.path.join(pg, "proteingym", "merge.py"))
    ok, why = _run_script_main(merge, ["merge.py", "--DMS_assays_location", args.dms_folder, "--model_scores_location", scores, "--DMS_reference_file", sub_csv,
                                       "--config_file", cfg_path])
    if not ok:
        print(f"accept_real_weights: the checkout's merge.py failed on the CSVs written here: {why}", file=sys.stderr)
        return 1
    perf = _load_script("pg_accept_performance", os.path.join(pg, "proteingym", "performance_DMS_benchmarks.py"))
    if patch_performance:
        patch_performance(perf)
    perf_dir = os.path.join(args.out, "performance")
    ok, why = _run_script_main(perf, ["performance_DMS_benchmarks.py", "--input_scoring_files_folder", os.path.join(scores, "merged_scores"),
                                      "--output_performance_file_folder", perf_dir, "--DMS_reference_file_path", sub_csv, "--DMS_data_folder", args.dms_folder,
                                      "--config_file", cfg_path])
    table_path = os.path.join(perf_dir, "Spearman", "DMS_substitutions_Spearman_DMS_level.csv")
    if not os.path.exists(table_path):
        print(f"accept_real_weights: performance_DMS_benchmarks.py wrote no per-assay Spearman table ({why})", file=sys.stderr)
        return 1
    if not ok:
        print(f"(performance_DMS_benchmarks.py stopped after its per-assay tables -- {why} -- as it does on a subset without every taxon / depth class)")
    table = pd.read_csv(table_path, index_col="DMS ID")
    published = pd.read_csv(args.published or os.path.join(pg, PUBLISHED), index_col="DMS ID")

    from scipy.stats import spearmanr
    rows, worst = [], 0.0
    for dms_id in subset["DMS_id"]:
        merged = pd.read_csv(os.path.join(scores, "merged_scores", f"{dms_id}.csv"))
        for key in models:
            col = clean.get(key, key)
            ours = float(table.loc[dms_id, col])
            own = float(spearmanr(merged["DMS_score"], merged[key])[0])
            want = float(published.loc[dms_id, col]) if dms_id in published.index and col in published.columns else float("nan")
            diff = abs(ours - want) if want == want else float("nan")
            worst = max(worst, diff) if diff == diff else worst
            rows.append(dict(DMS_id=dms_id, model=col, published=want, reference_script_on_our_scores=ours, scipy_on_our_scores=round(own, 6),
                             abs_difference=diff, verdict="n/a" if diff != diff else ("ok" if diff <= args.tolerance + 1e-12 else "DIFFERS")))
    report = pd.DataFrame(rows)
    report.to_csv(os.path.join(args.out, "acceptance_report.csv"), index=False)
    print(report.to_string(index=False))
    compared = report[report["verdict"] != "n/a"]
    if compared.empty:
        print("accept_real_weights: the published table has none of these (assay, model) pairs", file=sys.stderr)
        return 1
    bad = compared[compared["verdict"] == "DIFFERS"]
    print(f"\n{len(compared) - len(bad)} of {len(compared)} values within {args.tolerance} of the published table (largest difference {worst:.4f}); "
          f"report: {os.path.join(args.out, 'acceptance_report.csv')}")
    return 1 if len(bad) else 0


if __name__ == "__main__":
    sys.exit(main())
