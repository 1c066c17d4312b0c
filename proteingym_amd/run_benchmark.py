"""Score many DMS assays with one or more ESM checkpoints on all GPUs of a node.

    python -m torch.distributed.run --nproc-per-node 8 -m proteingym_amd.run_benchmark \\
        --model-location esm1v_1.pt ... esm1v_5.pt --model_type ESM1v \\
        --dms_mapping reference_files/DMS_substitutions.csv --dms-input DMS_ProteinGym_substitutions \\
        --dms-output scores/ESM1v [--dms_indices 0 1 2 ...]

One process per GPU.  The (assay) work list is LPT-balanced over ranks by algorithmic FLOPs; every
rank scores its assays with every checkpoint (weights replicated: 2.6 GB per checkpoint), then
one RCCL all_gather moves the per-mutant score vectors to all ranks and rank 0 writes one
``<DMS_id>.csv`` per assay with exactly the columns the reference CLI writes
(/root/reference/proteingym/baselines/esm/compute_fitness.py:505,532-543).  Existing CSVs that
already hold every requested column are skipped (resume), unless --overwrite-prior-scores.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import pandas as pd

from . import dist as pdist
from . import esm as pesm


def create_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--model-location", type=str, nargs="+", required=True)
    p.add_argument("--model_type", type=str, nargs="+", default=["ESM1v"])
    p.add_argument("--dms_mapping", type=str, required=True)
    p.add_argument("--dms-input", type=str, required=True)
    p.add_argument("--dms-output", type=str, required=True)
    p.add_argument("--dms_indices", type=int, nargs="*", default=None, help="default: every row of the mapping")
    p.add_argument("--mutation-col", type=str, default="mutant")
    p.add_argument("--precision", type=str, default="f16x3", choices=sorted(pesm._lib.PRECISIONS))
    p.add_argument("--scoring-strategy", type=str, default="masked-marginals", choices=["masked-marginals", "wt-marginals"],
                   help="wt-marginals: one unmasked forward per assay (blended 1024-token windows with --scoring-window overlapping), the "
                        "ESM-1b launchers' strategy (scripts/scoring_DMS_zero_shot/scoring_ESM1b_substitutions.sh, "
                        "scripts/scoring_clinical_zero_shot/scoring_ESM1b_substitutions.sh: 2 525 genes) with the checkpoint resident "
                        "instead of re-read per gene")
    p.add_argument("--scoring-window", type=str, default="optimal", choices=["optimal", "overlapping"])
    p.add_argument("--offset-idx", type=int, default=1, help="first residue number when the mapping has no start_idx column")
    p.add_argument("--all-positions", action="store_true")
    p.add_argument("--overwrite-prior-scores", action="store_true")
    p.add_argument("--backend", type=str, default=None, help="torch.distributed backend (default nccl)")
    p.add_argument("--shard", type=str, default="assay", choices=["assay", "positions"],
                   help="unit of work given to a GPU: whole assays (default; enough for the 217-assay benchmark) or "
                        "chunks of masked positions inside assays (few, unequal assays: the log-prob tables are "
                        "gathered instead of the score vectors)")
    p.add_argument("--chunk-forwards", type=int, default=64, help="positions per work item with --shard positions")
    p.add_argument("--batch-short-tokens", type=int, default=200,
                   help="masked-marginals, --shard assay: assays of at most this many tokens (residues + 2) and at most --batch-short-rows "
                        "mutants are scored several at a time -- their masked copies share one launch sequence, each padded to the group's "
                        "longest member behind a key mask (a 40-residue protein alone is 42 x 42 rows: a sixth of one round of GEMM tiles); "
                        "same bits as one assay at a time; 0 = off")
    p.add_argument("--batch-short-rows", type=int, default=20000)
    p.add_argument("--batch-group-rows", type=int, default=98304, help="cap on masked copies x padded tokens of one group")
    p.add_argument("--write", type=str, default="owner", choices=["owner", "rank0"],
                   help="--shard assay with N > 1: 'owner' = every rank writes the CSVs of the assays it scored (parallel "
                        "I/O; rank 0 still receives every score vector through the all_gather and writes "
                        "scores_summary.csv), 'rank0' = rank 0 writes every CSV from the gathered vectors")
    return p


def column_names(model_locations, model_type):
    cols = [m.split("/")[-1].split(".")[0] for m in model_locations]
    return cols, (["Ensemble_ESM1v"] if "ESM1v" in model_type else [])


GPU_FLOPS_PER_S = 3.0e14          # planning constants only (ratios matter): sustained algorithmic rate of one GPU ...
HOST_S_PER_ROW = 4.0e-6           # ... and parse + CSV seconds per mutant row on the rank that owns the assay


WT_MARGINALS_FIXED_S = 3.0e-3     # launch-bound floor of one unmasked forward + table download


def wt_marginals_windows(n_tok: int, scoring_window: str) -> int:
    """Forwards wt-marginals runs for a protein of n_tok tokens (compute_fitness.py:433-475): one, or with 'overlapping'
    above 1024 tokens the windows ``compute_fitness.overlapping_windows`` lists."""
    if n_tok <= 1024 or scoring_window != "overlapping":
        return 1
    from .compute_fitness import overlapping_windows
    return len(overlapping_windows(n_tok))


def assay_seconds(seq_len: int, n_rows: int, n_checkpoints: int, strategy: str = "masked-marginals",
                  scoring_window: str = "optimal") -> float:
    """Planned wall time of one assay on one rank: the strategy's FLOPs for every checkpoint + the host work that
    scales with the number of mutants (parsing, CSV): the 537k-row assay costs 2 s of host time against 0.3 s of GPU."""
    if strategy == "wt-marginals":
        n_tok = seq_len + 2
        gpu = wt_marginals_windows(n_tok, scoring_window) * pdist.forward_flops(min(n_tok, 1024)) / GPU_FLOPS_PER_S + WT_MARGINALS_FIXED_S
        return gpu * n_checkpoints + n_rows * HOST_S_PER_ROW
    return pdist.assay_cost(seq_len) * n_checkpoints / GPU_FLOPS_PER_S + n_rows * HOST_S_PER_ROW


def plan_assays(mapping, todo, world, n_checkpoints, strategy: str = "masked-marginals", scoring_window: str = "optimal"):
    rows = mapping["DMS_total_number_mutants"] if "DMS_total_number_mutants" in mapping.columns else None   # (the clinical table has none)
    costs = [assay_seconds(len(str(mapping.iloc[i]["target_seq"])), int(rows.iloc[i]) if rows is not None and rows.iloc[i] == rows.iloc[i] else 0,
                           n_checkpoints, strategy, scoring_window) for i in todo]
    return pdist.lpt_partition(costs, world)


def plan_short_groups(candidates, n_tok, group_rows: int):
    """Groups of short assays that share a launch sequence.  ``candidates``: assay ids; ``n_tok[i]``: tokens of assay i.  Sorted by
    length so that the padding to the longest member stays small; a group is closed when (masked copies) x (padded tokens) --
    bounded from above by every position of every member -- would pass ``group_rows``.  Groups of one are not groups."""
    groups, cur, copies = [], [], 0
    for i in sorted(candidates, key=lambda i: (n_tok[i], i)):
        if cur and (copies + n_tok[i]) * n_tok[i] > group_rows:
            groups.append(cur)
            cur, copies = [], 0
        cur.append(i)
        copies += n_tok[i]
    if cur:
        groups.append(cur)
    return [g for g in groups if len(g) > 1]


class _DeviceScorer:
    """One checkpoint on this rank's GPU: score(seq, mutants, offset) = Assay.run()."""

    def __init__(self, location, device, precision, all_positions):
        self.model, self.alphabet = pesm.load_model_and_alphabet(location, device=device, precision=precision)
        self.all_positions = all_positions
        self.create_s = self.run_s = 0.0            # wall clock in Assay() (parse + upload) / Assay.run() (GPU + 8 bytes per mutant back)
        self.log = []                               # per assay: residues, rows, masked positions run, T, seconds in Assay() / Assay.run()

    def score(self, seq, mutants, offset):
        t0 = time.time()
        assay = pesm.Assay(self.model, seq, mutants, offset_idx=int(offset), alphabet=self.alphabet,
                           all_positions=self.all_positions)
        t1 = time.time()
        out = assay.run()
        t2 = time.time()
        self.log.append(dict(seq_len=len(seq), rows=len(mutants), positions_run=int(len(assay.positions)), T=int(assay.T),
                             create_s=t1 - t0, run_s=t2 - t1))
        assay.close()
        self.create_s += t1 - t0
        self.run_s += t2 - t1
        return out

    def score_group(self, assays):
        """[(seq, mutants, offset), ...] of SHORT assays -> their score vectors, from ONE launch sequence: every masked copy of every
        member is a row of one [copies, T] token matrix (T = the longest member; shorter ones end in <pad>, which the attention
        masks as keys and the position / token-dropout arithmetic does not count: pgmi_masked_logprobs), the table rows come
        back together and each member's mutants are looked up on the host (esm.score_parsed: the arithmetic of
        score_mutants_kernel).  Bits equal score() member by member (tests/test_gpu_cli.py)."""
        t0 = time.time()
        conv = self.alphabet.get_batch_converter()
        members = []
        for seq, mutants, offset in assays:
            _, _, toks = conv([("protein1", seq)])
            wt = np.asarray(toks[0], dtype=np.int32)
            parsed = pesm.parse_mutants(mutants, seq, int(offset))
            pos = np.arange(wt.size, dtype=np.int32) if self.all_positions else np.unique(parsed[0]).astype(np.int32)
            members.append((wt, parsed, pos))
        T = max(wt.size for wt, _, _ in members)
        copies = sum(pos.size for _, _, pos in members)
        tokens = np.full((copies, T), self.alphabet.padding_idx, dtype=np.int32)
        mask_pos = np.empty(copies, dtype=np.int32)
        b = 0
        for wt, _, pos in members:
            tokens[b:b + pos.size, :wt.size] = wt
            mask_pos[b:b + pos.size] = pos
            b += pos.size
        t1 = time.time()
        rows = self.model.masked_logprobs(tokens, mask_pos) if copies else np.zeros((0, 33), dtype=np.float32)
        t2 = time.time()
        outs, b = [], 0
        for wt, parsed, pos in members:
            table = np.full((wt.size, rows.shape[1]), np.nan, dtype=np.float32)
            table[pos] = rows[b:b + pos.size]
            b += pos.size
            outs.append(pesm.score_parsed(table, *parsed))
        t3 = time.time()
        work = [pos.size * pdist.forward_flops(wt.size) for wt, _, pos in members]
        for (seq, mutants, _), (wt, _, pos), w in zip(assays, members, work):
            share = w / max(sum(work), 1e-30)                  # the group's seconds, split by the members' algorithmic FLOPs
            self.log.append(dict(seq_len=len(seq), rows=len(mutants), positions_run=int(pos.size), T=int(wt.size),
                                 create_s=(t1 - t0 + t3 - t2) * share, run_s=(t2 - t1) * share, group_of=len(assays), padded_T=int(T)))
        self.create_s += t1 - t0 + t3 - t2
        self.run_s += t2 - t1
        return outs

    def close(self):
        self.model.close()


class _DeviceWtMarginals:
    """wt-marginals with a resident checkpoint: score(seq, mutants, offset) = one unmasked forward (or the blended
    overlapping windows) + label_row for every mutant (compute_fitness.py:433-485).  The table comes from the same
    function the single-assay CLI calls; the look-ups are ``esm.score_from_table`` (f32 difference, double
    accumulation in mutation order: the arithmetic of the reference's python loop)."""

    def __init__(self, location, device, precision, scoring_window):
        self.model, self.alphabet = pesm.load_model_and_alphabet(location, device=device, precision=precision)
        self.scoring_window = scoring_window
        self.create_s = self.run_s = 0.0
        self.log = []

    def score(self, seq, mutants, offset):
        from .compute_fitness import wt_marginals_table
        t0 = time.time()
        table = wt_marginals_table(self.model, self.alphabet, seq, self.scoring_window)
        t1 = time.time()
        out = pesm.score_from_table(table[0], mutants, seq, int(offset))
        t2 = time.time()
        n_tok = len(seq) + 2
        self.log.append(dict(seq_len=len(seq), rows=len(mutants), positions_run=wt_marginals_windows(n_tok, self.scoring_window),
                             T=min(n_tok, 1024), create_s=t2 - t1, run_s=t1 - t0))
        self.run_s += t1 - t0
        self.create_s += t2 - t1
        return out

    def close(self):
        self.model.close()


def _finish_frame(df, cols, ens_cols, vectors):
    """Add the checkpoint columns (+ the plain-mean ensemble, compute_fitness.py:532-537) to an assay's frame."""
    for c, name in enumerate(cols):
        df[name] = vectors[c]
    if ens_cols:
        df["Ensemble_ESM1v"] = 0.0
        for name in cols:
            df["Ensemble_ESM1v"] += df[name]
        df["Ensemble_ESM1v"] /= len(cols)
    return df


def _write_csv(df, path):
    df.to_csv(path + ".tmp", index=False)
    os.replace(path + ".tmp", path)


def main(args, make_model=None):
    """``make_model`` is a test seam: (location) -> object with score(seq, mutants, offset) and close()."""
    rank, local_rank, world = pdist.init_from_env(args.backend)
    mapping = pd.read_csv(args.dms_mapping)
    indices = list(range(len(mapping))) if args.dms_indices is None else list(args.dms_indices)
    cols, ens_cols = column_names(args.model_location, args.model_type)
    os.makedirs(args.dms_output, exist_ok=True)

    # rank 0 decides what is left to do and broadcasts it: every collective below is shaped by this list, so it must not
    # depend on each rank's own view of the output folder (NFS lag, an owner-writing rank finishing early)
    todo = []
    if rank == 0:
        for i in indices:
            row = mapping.iloc[i]
            out = os.path.join(args.dms_output, str(row["DMS_id"]) + ".csv")
            if os.path.exists(out) and not args.overwrite_prior_scores:
                have = pd.read_csv(out, nrows=0).columns
                if all(c in have for c in cols + ens_cols):
                    continue
            todo.append(i)
    if world > 1:
        import torch.distributed as tdist
        box = [todo]
        tdist.broadcast_object_list(box, src=0)
        todo = box[0]
    wt_marginals = args.scoring_strategy == "wt-marginals"
    if args.shard == "positions":
        if wt_marginals:
            raise SystemExit("run_benchmark: --shard positions cuts MASKED positions; wt-marginals has one forward per assay")
        return main_position_shards(args, mapping, todo, cols, ens_cols, rank, local_rank, world, make_model=make_model)
    if not wt_marginals and args.scoring_window == "overlapping":
        raise SystemExit("Overlapping not yet implemented for masked-marginals")        # compute_fitness.py:487-488
    assignment = plan_assays(mapping, todo, world, len(args.model_location), args.scoring_strategy, args.scoring_window)
    mine = [todo[k] for k in assignment[rank]]

    t0 = time.time()
    from concurrent.futures import ThreadPoolExecutor
    frames, local, reads = {}, {}, {}
    clock = {"wait_read_s": 0.0, "score_s": 0.0, "wait_write_s": 0.0}

    def read_frame(i):
        row = mapping.iloc[i].replace(np.nan, "")
        mutant_col = row["DMS_mutant_column"] if "DMS_mutant_column" in mapping.columns else args.mutation_col
        return (pd.read_csv(os.path.join(args.dms_input, row["DMS_filename"])), mutant_col, str(row["target_seq"]).upper(),
                row["start_idx"] if "start_idx" in mapping.columns and row["start_idx"] != "" else args.offset_idx)

    # the DMS files are read by a background thread in scoring order, ahead of the GPU (the C parser and the scoring calls
    # both drop the GIL); every frame is kept: the checkpoint columns are added to it at the end
    rows_of = {i: int(mapping.iloc[i]["DMS_total_number_mutants"]) if "DMS_total_number_mutants" in mapping.columns else 0 for i in mine}
    order = sorted(mine, key=lambda i: -rows_of[i])      # most rows first: the CSV written last, un-overlapped, is a small one
    # short assays with few rows go several at a time (score_group), after the others: they are the tail of `order` anyway
    groups = []
    if not wt_marginals and args.batch_short_tokens > 0:
        n_tok = {i: len(str(mapping.iloc[i]["target_seq"])) + 2 for i in mine}
        groups = plan_short_groups([i for i in mine if n_tok[i] <= args.batch_short_tokens and rows_of[i] <= args.batch_short_rows],
                                   n_tok, args.batch_group_rows)
    in_group = {i for g in groups for i in g}
    reader = ThreadPoolExecutor(max_workers=1)
    for i in [i for i in order if i not in in_group] + [i for g in groups for i in g]:
        reads[i] = reader.submit(read_frame, i)

    def frame(i):
        if i not in frames:
            t = time.time()
            frames[i] = reads.pop(i).result()
            clock["wait_read_s"] += time.time() - t
        return frames[i]

    owner_writes = args.write == "owner"
    # the owner's CSVs are written by a background thread as soon as an assay has its last checkpoint column: pandas
    # formatting (2.47 M rows over the benchmark) overlaps the scoring of the following assays (the C calls drop the GIL);
    writer = ThreadPoolExecutor(max_workers=2) if owner_writes else None
    pending = []
    assay_log = []
    for ci, loc in enumerate(args.model_location):
        t = time.time()
        model = make_model(loc) if make_model is not None else \
            (_DeviceWtMarginals(loc, local_rank, args.precision, args.scoring_window) if wt_marginals
             else _DeviceScorer(loc, local_rank, args.precision, args.all_positions))
        clock["checkpoint_load_s"] = clock.get("checkpoint_load_s", 0.0) + time.time() - t
        last = ci == len(args.model_location) - 1
        grouped = hasattr(model, "score_group") and bool(groups)
        logged = []
        for unit in ([[i] for i in order if i not in in_group] + groups) if grouped else [[i] for i in order]:
            fr = [frame(i) for i in unit]
            t = time.time()
            if len(unit) == 1:
                df, mutant_col, seq, offset = fr[0]
                got = [model.score(seq, [str(m) for m in df[mutant_col]], offset)]
            else:
                got = model.score_group([(seq, [str(m) for m in df[mutant_col]], offset) for df, mutant_col, seq, offset in fr])
            clock["score_s"] += time.time() - t
            logged += unit
            for i, v, (df, _, _, _) in zip(unit, got, fr):
                local.setdefault(i, []).append(np.asarray(v, dtype=np.float64))
                if last and writer is not None:
                    pending.append(writer.submit(_write_csv, _finish_frame(df, cols, ens_cols, local[i]),
                                                 os.path.join(args.dms_output, str(mapping.iloc[i]["DMS_id"]) + ".csv")))
        for k in ("create_s", "run_s"):
            if hasattr(model, k):
                clock["assay_" + k] = clock.get("assay_" + k, 0.0) + getattr(model, k)
        assay_log += [dict(e, checkpoint=ci, DMS_id=str(mapping.iloc[i]["DMS_id"])) for e, i in zip(getattr(model, "log", []), logged)]
        model.close()
    t = time.time()
    for f in pending:
        f.result()                                 # re-raises a writer's exception
    if writer is not None:
        writer.shutdown()
    reader.shutdown()
    clock["wait_write_s"] = time.time() - t
    # exchange: per item a [n_checkpoints * n_mut] vector
    sizes = []
    n_rows = {}
    for k, i in enumerate(todo):
        row = mapping.iloc[i]
        n = int(row["DMS_total_number_mutants"]) if "DMS_total_number_mutants" in mapping.columns and i not in frames \
            else (len(frames[i][0]) if i in frames else None)
        n_rows[i] = n
    if world > 1:
        import torch
        import torch.distributed as tdist
        cnt = torch.zeros(len(todo), dtype=torch.int64, device="cuda" if tdist.get_backend() == "nccl" else "cpu")
        for k, i in enumerate(todo):
            if i in frames:
                cnt[k] = len(frames[i][0])
        tdist.all_reduce(cnt)                       # exact row counts (files may differ from the mapping)
        for k, i in enumerate(todo):
            n_rows[i] = int(cnt[k])
    sizes = [n_rows[i] * len(cols) for i in todo]
    payload = {assignment[rank][j]: np.concatenate(local[i]) for j, i in enumerate(mine)}
    dev = None
    if world > 1:
        import torch.distributed as tdist
        dev = "cuda" if tdist.get_backend() == "nccl" else "cpu"
    allv = pdist.gather_score_vectors(payload, sizes, assignment, device=dev)

    if rank == 0:
        n_mut = 0
        summary = []
        for k, i in enumerate(todo):
            row = mapping.iloc[i].replace(np.nan, "")
            v = allv[k].reshape(len(cols), -1)
            if not owner_writes:
                df = frames[i][0] if i in frames else pd.read_csv(os.path.join(args.dms_input, row["DMS_filename"]))
                _write_csv(_finish_frame(df, cols, ens_cols, v), os.path.join(args.dms_output, str(row["DMS_id"]) + ".csv"))
            n_mut += v.shape[1]
            summary.append({"DMS_id": str(row["DMS_id"]), "mutants": v.shape[1],
                            **{f"mean_{name}": float(np.mean(v[c])) if v.shape[1] else float("nan") for c, name in enumerate(cols)}})
        if world > 1:                              # what the gathered vectors are for when the owners write the CSVs
            _write_csv(pd.DataFrame(summary), os.path.join(args.dms_output, "scores_summary.csv"))
        dt = time.time() - t0
        print(f"scored {len(todo)} assays / {n_mut} mutants x {len(cols)} checkpoint(s) on {world} GPU(s) "
              f"in {dt:.1f}s = {n_mut / max(dt, 1e-9):.1f} mutants/s (ensemble rate)")
        print("rank 0 wall clock: " + ", ".join(f"{k} {v:.1f}" for k, v in clock.items())
              + " (score_s = mutant parse + upload + GPU; checkpoint load and the rest are the difference)")
        stats = dict(seconds=dt, mutants=n_mut, assays=len(todo), checkpoints=len(cols), world=world, rank0_wall_clock=dict(clock),
                     rank0_assays=assay_log)
    else:
        stats = None
    if world > 1:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()
    return stats


def main_position_shards(args, mapping, todo, cols, ens_cols, rank, local_rank, world, make_model=None):
    """--shard positions: every assay's masked positions are cut into chunks, chunks are LPT-balanced over the ranks,
    each rank fills the table rows of its chunks, ONE all_gather moves the partial tables (NaN = not mine), every
    rank merges them and rank 0 scores the mutants from the complete tables (bit-identical to Assay.run()) and
    writes the CSVs.  ``make_model`` is a test seam: (location) -> object with table_rows(seq, positions, offset)."""
    t0 = time.time()
    frames = []
    for i in todo:
        row = mapping.iloc[i].replace(np.nan, "")
        mutant_col = row["DMS_mutant_column"] if "DMS_mutant_column" in mapping.columns else args.mutation_col
        df = pd.read_csv(os.path.join(args.dms_input, row["DMS_filename"]))
        seq = str(row["target_seq"]).upper()
        offset = int(row["start_idx"]) if "start_idx" in mapping.columns and row["start_idx"] != "" else 1
        muts = [str(m) for m in df[mutant_col]]
        pos = np.arange(len(seq) + 2, dtype=np.int32) if args.all_positions else pesm.positions_read(muts, seq, offset)
        frames.append(dict(df=df, seq=seq, offset=offset, mutants=muts, positions=pos, dms_id=str(row["DMS_id"])))
    items, assignment = pdist.plan_position_chunks([len(f["seq"]) for f in frames], [f["positions"] for f in frames],
                                                   world, chunk_forwards=args.chunk_forwards)
    n_toks = [len(f["seq"]) + 2 for f in frames]
    dev = None
    if world > 1:
        import torch.distributed as tdist
        dev = "cuda" if tdist.get_backend() == "nccl" else "cpu"
    for ci, loc in enumerate(args.model_location):
        if make_model is not None:
            model = make_model(loc)
        else:
            model = _DeviceTables(loc, local_rank, args.precision)
        local = {a: np.full((n_toks[a], 33), np.nan, dtype=np.float32) for a in range(len(frames))}
        for k in assignment[rank]:
            a, chunk = items[k]
            rows = model.table_rows(frames[a]["seq"], chunk, frames[a]["offset"])
            local[a][chunk] = rows
        tables = pdist.gather_tables(local, n_toks, device=dev)
        if hasattr(model, "close"):
            model.close()
        if rank == 0:
            for a, f in enumerate(frames):
                f["df"][cols[ci]] = pesm.score_from_table(tables[a], f["mutants"], f["seq"], f["offset"])
    if rank == 0:
        n_mut = 0
        for f in frames:
            df = f["df"]
            if ens_cols:
                df["Ensemble_ESM1v"] = sum(df[c] for c in cols) / len(cols)
            out = os.path.join(args.dms_output, f["dms_id"] + ".csv")
            df.to_csv(out + ".tmp", index=False)
            os.replace(out + ".tmp", out)
            n_mut += len(df)
        dt = time.time() - t0
        print(f"scored {len(frames)} assays / {n_mut} mutants x {len(cols)} checkpoint(s) on {world} GPU(s), "
              f"{len(items)} position chunks, in {dt:.1f}s = {n_mut / max(dt, 1e-9):.1f} mutants/s (ensemble rate)")
    if world > 1:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()


class _DeviceTables:
    """table_rows() on the GPU: an Assay restricted to the chunk's positions (the alignment of work items to
    masked positions is the device path's own: windows, chunking and the head run as in pgmi_assay_run)."""

    def __init__(self, location, device, precision):
        self.model, self.alphabet = pesm.load_model_and_alphabet(location, device=device, precision=precision)

    def table_rows(self, seq, positions, offset):
        first = seq[0] + str(offset) + ("A" if seq[0] != "A" else "C")          # any valid mutant: only the table is read
        a = pesm.Assay(self.model, seq, [first], offset_idx=offset, alphabet=self.alphabet, positions=positions)
        _, table = a.run(want_table=True)
        a.close()
        return table[np.asarray(positions)]

    def close(self):
        self.model.close()


if __name__ == "__main__":
    main(create_parser().parse_args())
