"""Score many assays with the Tranception or MSA Transformer path on all GPUs of a node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m proteingym_amd.run_sharded tranception [--indices 0 1 2 ...] -- \\
        --checkpoint Tranception_Large --DMS_reference_file_path reference_files/DMS_substitutions.csv \\
        --DMS_data_folder DMS_ProteinGym_substitutions --output_scores_folder scores/Tranception \\
        --inference_time_retrieval --MSA_folder MSA_files --MSA_weights_folder MSA_weights
    ... run_sharded msa_transformer -- --model-location esm_msa1b.pt --model_type MSA_transformer \\
        --dms_mapping reference_files/DMS_substitutions.csv --dms-input ... --dms-output ... --msa-path ... --seeds 1 2 3 4 5

Everything after ``--`` is the single-assay command line of the reference launcher
(scripts/scoring_DMS_zero_shot/scoring_Tranception_substitutions.sh, scoring_MSA_transformer_substitutions.sh)
WITHOUT its assay index.  One process per GPU; assays are independent units, LPT-balanced over the ranks
by an algorithmic cost estimate, every rank writes the CSVs of its own assays: no data-path collective,
one barrier at the end ("Tranception and pseudo-ppl shard the same way").
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import pandas as pd

from . import dist as pdist

BASELINES = {
    # name: (module with create_parser()/main(args), index flag, reference-file flag, device flag)
    "tranception": ("proteingym_amd.score_tranception_proteingym", "--DMS_index", "--DMS_reference_file_path", "--device"),
    "msa_transformer": ("proteingym_amd.compute_fitness", "--dms_index", "--dms_mapping", "--device"),
}


def assay_cost(baseline: str, row) -> float:
    """Relative algorithmic cost of one assay (only ratios matter)."""
    L = len(str(row["target_seq"]))
    n_mut = float(row["DMS_total_number_mutants"]) if "DMS_total_number_mutants" in row and row["DMS_total_number_mutants"] == row["DMS_total_number_mutants"] else 1000.0
    T = min(L + 2, 1024)
    if baseline == "tranception":
        return (n_mut + 1.0) * T * (1.0 + T / 7680.0)          # tokens x (linear + attention share) per scored sequence
    if "MSA_start" in row and "MSA_end" in row and row["MSA_start"] == row["MSA_start"]:
        T = min(int(row["MSA_end"]) - int(row["MSA_start"]) + 2, 1024)
    return float(T) * T                                         # (masked columns) x (tokens per forward ~ rows x T)


def _value_after(argv, flag):
    for i, a in enumerate(argv):
        if a == flag and i + 1 < len(argv):
            return argv[i + 1]
    raise SystemExit(f"run_sharded: the single-assay arguments must contain {flag}")


def plan(baseline: str, mapping: pd.DataFrame, indices, world: int):
    costs = [assay_cost(baseline, mapping.iloc[i]) for i in indices]
    assignment = pdist.lpt_partition(costs, world)
    return [[indices[k] for k in part] for part in assignment]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if "--" not in argv:
        raise SystemExit(__doc__)
    cut = argv.index("--")
    ap = argparse.ArgumentParser()
    ap.add_argument("baseline", choices=sorted(BASELINES))
    ap.add_argument("--indices", type=int, nargs="*", default=None, help="default: every row of the reference file")
    ap.add_argument("--dry-run", action="store_true", help="print this rank's assays and exit")
    ap.add_argument("--shard", choices=["assay", "positions"], default="assay",
                    help="msa_transformer only: 'positions' = every rank works on EVERY assay and forwards its share of the (seed, "
                         "masked position) pairs; the log-prob tables are all_gathered (one BLAT-size assay x 5 seeds is ~190 s "
                         "of serial work: sharding whole assays cannot balance a handful of them)")
    ap.add_argument("--backend", type=str, default=None)
    own = ap.parse_args(argv[:cut])
    rest = argv[cut + 1:]
    module, index_flag, ref_flag, device_flag = BASELINES[own.baseline]
    if index_flag in rest:
        raise SystemExit(f"run_sharded: do not pass {index_flag}; assays are distributed over the ranks")
    rank, local_rank, world = pdist.init_from_env(own.backend)
    mapping = pd.read_csv(_value_after(rest, ref_flag))
    indices = list(range(len(mapping))) if own.indices is None else list(own.indices)
    by_position = own.shard == "positions"
    if by_position and own.baseline != "msa_transformer":
        raise SystemExit("run_sharded: --shard positions is implemented for msa_transformer")
    mine = list(indices) if by_position else plan(own.baseline, mapping, indices, world)[rank]
    if by_position:
        rest = rest + ["--shard-positions"]
    print(f"[rank {rank}/{world}] {own.baseline}: assays {mine}", flush=True)
    if not own.dry_run:
        import importlib
        mod = importlib.import_module(module)
        t0 = time.time()
        failed = []
        for i in mine:
            args = mod.create_parser().parse_args(rest + [index_flag, str(i), device_flag, str(local_rank)])
            try:                                   # one assay's failure (a missing MSA, sys.exit in the single-assay CLI ...) must
                mod.main(args)                     # neither skip this rank's other assays nor leave the peers in the barrier
            except BaseException as e:
                if isinstance(e, KeyboardInterrupt):
                    raise
                failed.append((i, f"{type(e).__name__}: {e}"))
                print(f"[rank {rank}] assay {i} FAILED: {type(e).__name__}: {e}", flush=True)
        print(f"[rank {rank}] {len(mine) - len(failed)} of {len(mine)} assays in {time.time() - t0:.1f}s", flush=True)
    else:
        failed = []
    n_failed = len(failed)
    if world > 1:
        import torch
        import torch.distributed as tdist
        cnt = torch.tensor([n_failed], dtype=torch.int64, device="cuda" if tdist.get_backend() == "nccl" else "cpu")
        tdist.all_reduce(cnt)                      # doubles as the final barrier
        n_failed = int(cnt.item())
        tdist.destroy_process_group()
    if n_failed:
        for i, why in failed:
            print(f"[rank {rank}] failed assay {i}: {why}", file=sys.stderr, flush=True)
        raise SystemExit(f"run_sharded: {n_failed} assay(s) failed (see the per-rank messages)")
    return mine


if __name__ == "__main__":
    main()
