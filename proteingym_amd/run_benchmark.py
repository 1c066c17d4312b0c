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
        try:
            out = assay.run()
        finally:
            assay.close()
        t2 = time.time()
        self.log.append(dict(seq_len=len(seq), rows=len(mutants), positions_run=int(len(assay.positions)), T=int(assay.T),
                             create_s=t1 - t0, run_s=t2 - t1))
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


# ---- failure isolation ------------------------------------------------------------------------------------------------------
# The reference isolates failures for free: one process per --dms_index (scripts/scoring_DMS_zero_shot/
# scoring_ESM1v_substitutions.sh:21-31), so a wild-type mismatch in one DMS file (the assertion of compute_fitness.py:243)
# loses that assay only.  Here many assays share a process AND a collective: a unit that fails fills NaN, every rank still
# reaches every collective, the failed assay's CSV is not written (a re-run picks it up), the others are written as in a clean
# run, and the job exits non-zero with a summary.  PGMI_EOVERFLOW (a 16-bit mode met an activation beyond fp16's range) is not
# a failure: that (assay, checkpoint) is re-scored on an fp32 model of the same checkpoint, built once per rank when first
# needed, and scores_summary.csv says so (precision_<checkpoint> = fp32).
def is_overflow(e) -> bool:
    return isinstance(e, pesm.PgmiError) and getattr(e, "code", None) == pesm._lib.EOVERFLOW


def _describe(e) -> str:
    return f"{type(e).__name__}: {e}"


def _seam(make_model, loc, precision=None):
    """The test seam: ``make_model(location)``; one that also takes ``precision=`` can serve the fp32 retry."""
    if precision is None:
        return make_model(loc)
    import inspect
    try:
        takes = "precision" in inspect.signature(make_model).parameters
    except (TypeError, ValueError):
        takes = False
    return make_model(loc, precision=precision) if takes else None


class _Fp32Retry:
    """The fp32 model of one checkpoint on this rank, built when the first unit overflows and closed with the checkpoint."""

    def __init__(self, build, requested: str):
        self.build, self.requested, self.model = build, requested, None

    def wanted(self, e) -> bool:
        return is_overflow(e) and self.requested != "fp32"

    def get(self):
        if self.model is None:
            self.model = self.build()
            if self.model is None:
                raise pesm.PgmiError("no fp32 model available for the retry")
        return self.model

    def close(self):
        if self.model is not None and hasattr(self.model, "close"):
            self.model.close()
        self.model = None


def _exchange_reports(world, *dicts):
    """Union of per-rank {key -> value} reports (failed assays, fp32 retries) on every rank; the lower rank's value wins."""
    if world == 1:
        return [dict(d) for d in dicts]
    import torch.distributed as tdist
    got = [None] * world
    tdist.all_gather_object(got, [dict(d) for d in dicts])
    out = [{} for _ in dicts]
    for per_rank in got:
        for o, d in zip(out, per_rank):
            for k, v in d.items():
                o.setdefault(k, v)
    return out


def _summary_rows(mapping, todo, cols, vectors, failed, used):
    """scores_summary.csv: one row per assay of the job -- rows, mean per checkpoint column, status, the precision every
    (assay, checkpoint) was scored in, and the message of a failed one."""
    rows = []
    for k, i in enumerate(todo):
        v = vectors[k]
        ok = i not in failed
        rows.append({"DMS_id": str(mapping.iloc[i]["DMS_id"]), "mutants": int(v.shape[1]),
                     **{f"mean_{name}": float(np.mean(v[c])) if v.shape[1] and ok else float("nan") for c, name in enumerate(cols)},
                     "status": "ok" if ok else "failed",
                     **{f"precision_{name}": used.get((i, c), "") for c, name in enumerate(cols)},
                     "error": failed.get(i, "")})
    return rows


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
    failed, used = {}, {}                           # assay -> message;  (assay, checkpoint index) -> precision of a retry
    unread = {}

    def frame_or_none(i):                           # an unreadable DMS file fails that assay (once), not the rank
        if i in unread:
            return None
        try:
            return frame(i)
        except BaseException as e:                  # noqa: BLE001 -- isolation: see the note above _Fp32Retry
            if isinstance(e, KeyboardInterrupt):
                raise
            reads.pop(i, None)
            unread[i] = _describe(e)
            failed.setdefault(i, unread[i])
            print(f"[rank {rank}] assay {mapping.iloc[i]['DMS_id']} FAILED: {unread[i]}", flush=True)
            return None

    def blank(i):
        return np.full(len(frames[i][0]), np.nan) if i in frames else np.zeros(0)

    def real_model(loc, precision):
        return (_DeviceWtMarginals(loc, local_rank, precision, args.scoring_window) if wt_marginals
                else _DeviceScorer(loc, local_rank, precision, args.all_positions))

    for ci, loc in enumerate(args.model_location):
        t = time.time()
        try:
            model = _seam(make_model, loc) if make_model is not None else real_model(loc, args.precision)
        except BaseException as e:                  # noqa: BLE001 -- a checkpoint this rank cannot load fails this rank's assays
            if isinstance(e, KeyboardInterrupt):
                raise
            print(f"[rank {rank}] checkpoint {loc} FAILED to load: {_describe(e)}", flush=True)
            for i in mine:
                frame_or_none(i)
                failed.setdefault(i, f"checkpoint {loc}: {_describe(e)}")
                local.setdefault(i, []).append(blank(i))
            continue
        retry = _Fp32Retry((lambda loc=loc: _seam(make_model, loc, "fp32")) if make_model is not None
                           else (lambda loc=loc: real_model(loc, "fp32")), args.precision)
        clock["checkpoint_load_s"] = clock.get("checkpoint_load_s", 0.0) + time.time() - t
        last = ci == len(args.model_location) - 1
        grouped = hasattr(model, "score_group") and bool(groups)
        logged = []

        def score_one(i, fr):
            """One assay on this checkpoint -> its vector (NaN when it failed, now or on an earlier checkpoint)."""
            if i in failed:
                return blank(i)
            df, mutant_col, seq, offset = fr
            try:
                muts = [str(m) for m in df[mutant_col]]
                try:
                    return np.asarray(model.score(seq, muts, offset), dtype=np.float64)
                except pesm.PgmiError as e:
                    if not retry.wanted(e):
                        raise
                    print(f"[rank {rank}] assay {mapping.iloc[i]['DMS_id']} x {cols[ci]}: {args.precision} left the fp16 range, "
                          "re-scoring this assay in fp32", flush=True)
                    v = np.asarray(retry.get().score(seq, muts, offset), dtype=np.float64)
                    used[(i, ci)] = "fp32"
                    return v
            except BaseException as e:              # noqa: BLE001
                if isinstance(e, KeyboardInterrupt):
                    raise
                failed[i] = _describe(e)
                print(f"[rank {rank}] assay {mapping.iloc[i]['DMS_id']} x {cols[ci]} FAILED: {failed[i]}", flush=True)
                return blank(i)

        for unit in ([[i] for i in order if i not in in_group] + groups) if grouped else [[i] for i in order]:
            fr = [frame_or_none(i) for i in unit]
            t = time.time()
            got = None
            if len(unit) > 1 and not any(i in failed for i in unit):
                try:
                    got = [np.asarray(v, dtype=np.float64) for v in
                           model.score_group([(seq, [str(m) for m in df[mutant_col]], offset) for df, mutant_col, seq, offset in fr])]
                except BaseException as e:          # noqa: BLE001 -- which member it was shows one at a time
                    if isinstance(e, KeyboardInterrupt):
                        raise
                    got = None
            if got is None:
                got = [score_one(i, f) for i, f in zip(unit, fr)]
            clock["score_s"] += time.time() - t
            logged += [i for i in unit if i not in failed and (i, ci) not in used]
            for i, v in zip(unit, got):
                local.setdefault(i, []).append(v)
                if last and writer is not None and i not in failed:
                    pending.append((i, writer.submit(_write_csv, _finish_frame(frames[i][0], cols, ens_cols, local[i]),
                                                     os.path.join(args.dms_output, str(mapping.iloc[i]["DMS_id"]) + ".csv"))))
        for k in ("create_s", "run_s"):
            if hasattr(model, k):
                clock["assay_" + k] = clock.get("assay_" + k, 0.0) + getattr(model, k)
        if len(getattr(model, "log", [])) == len(logged):
            assay_log += [dict(e, checkpoint=ci, DMS_id=str(mapping.iloc[i]["DMS_id"])) for e, i in zip(model.log, logged)]
        model.close()
        retry.close()
    t = time.time()
    for i, f in pending:
        try:
            f.result()
        except BaseException as e:                  # noqa: BLE001 -- a CSV that could not be written is a failed assay
            if isinstance(e, KeyboardInterrupt):
                raise
            failed[i] = "writing the CSV: " + _describe(e)
            print(f"[rank {rank}] assay {mapping.iloc[i]['DMS_id']} FAILED: {failed[i]}", flush=True)
    if writer is not None:
        writer.shutdown()
    reader.shutdown()
    clock["wait_write_s"] = time.time() - t
    # exchange: per item a [n_checkpoints * n_mut] vector
    n_rows = {i: (len(frames[i][0]) if i in frames else 0) for i in todo}
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
    failed, used = _exchange_reports(world, failed, used)
    sizes = [n_rows[i] * len(cols) for i in todo]
    payload = {assignment[rank][j]: np.concatenate(local[i]) if local.get(i) else np.zeros(0) for j, i in enumerate(mine)}
    dev = None
    if world > 1:
        import torch.distributed as tdist
        dev = "cuda" if tdist.get_backend() == "nccl" else "cpu"
    allv = pdist.gather_score_vectors(payload, sizes, assignment, device=dev)

    if rank == 0:
        n_mut = 0
        vectors = []
        for k, i in enumerate(todo):
            row = mapping.iloc[i].replace(np.nan, "")
            v = allv[k].reshape(len(cols), -1)
            vectors.append(v)
            if i in failed:
                continue
            if not owner_writes:
                try:
                    df = frames[i][0] if i in frames else pd.read_csv(os.path.join(args.dms_input, row["DMS_filename"]))
                    _write_csv(_finish_frame(df, cols, ens_cols, v), os.path.join(args.dms_output, str(row["DMS_id"]) + ".csv"))
                except BaseException as e:          # noqa: BLE001
                    if isinstance(e, KeyboardInterrupt):
                        raise
                    failed[i] = "writing the CSV: " + _describe(e)
                    continue
            n_mut += v.shape[1]
        if world > 1 or failed or used:            # what the gathered vectors are for when the owners write the CSVs
            _write_csv(pd.DataFrame(_summary_rows(mapping, todo, cols, vectors, failed, used)),
                       os.path.join(args.dms_output, "scores_summary.csv"))
        dt = time.time() - t0
        print(f"scored {len(todo) - len(failed)} assays / {n_mut} mutants x {len(cols)} checkpoint(s) on {world} GPU(s) "
              f"in {dt:.1f}s = {n_mut / max(dt, 1e-9):.1f} mutants/s (ensemble rate)")
        print("rank 0 wall clock: " + ", ".join(f"{k} {v:.1f}" for k, v in clock.items())
              + " (score_s = mutant parse + upload + GPU; checkpoint load and the rest are the difference)")
        stats = dict(seconds=dt, mutants=n_mut, assays=len(todo) - len(failed), checkpoints=len(cols), world=world,
                     rank0_wall_clock=dict(clock), rank0_assays=assay_log, failed=dict(failed),
                     precision_used={f"{mapping.iloc[i]['DMS_id']}:{cols[c]}": p for (i, c), p in used.items()})
    else:
        stats = None
    if world > 1:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()
    _exit_on_failures(mapping, failed, rank)
    return stats


def _exit_on_failures(mapping, failed, rank, who="run_benchmark"):
    if not failed:
        return
    if rank == 0:
        import sys
        for i, why in sorted(failed.items()):
            print(f"{who}: assay {mapping.iloc[i]['DMS_id']} failed: {why}", file=sys.stderr, flush=True)
    raise SystemExit(f"{who}: {len(failed)} assay(s) failed (scores_summary.csv lists them); their CSVs were not written")


def main_position_shards(args, mapping, todo, cols, ens_cols, rank, local_rank, world, make_model=None):
    """--shard positions: every assay's masked positions are cut into chunks, chunks are LPT-balanced over the ranks,
    each rank fills the table rows of its chunks, ONE all_gather moves the partial tables (NaN = not mine), every
    rank merges them and rank 0 scores the mutants from the complete tables (bit-identical to Assay.run()) and
    writes the CSVs.  ``make_model`` is a test seam: (location) -> object with table_rows(seq, positions, offset).

    Failures (see the note above ``_Fp32Retry``): an assay whose file or mutant column is bad anywhere is dropped on every
    rank BEFORE the plan (the plan must be the same everywhere); a chunk that fails on one rank fails its assay; a chunk
    that leaves the fp16 range sends the WHOLE assay (every rank's chunks of it) through fp32 models for that checkpoint,
    so a table never mixes precisions and the scores do not depend on the number of ranks.  The reports travel as
    objects between the passes; every rank takes part in every exchange."""
    t0 = time.time()
    frames, failed, used = [], {}, {}
    for a, i in enumerate(todo):
        row = mapping.iloc[i].replace(np.nan, "")
        seq = str(row["target_seq"]).upper()
        f = dict(df=None, seq=seq, offset=1, mutants=[], positions=np.zeros(0, np.int32), dms_id=str(row["DMS_id"]))
        try:
            mutant_col = row["DMS_mutant_column"] if "DMS_mutant_column" in mapping.columns else args.mutation_col
            f["df"] = pd.read_csv(os.path.join(args.dms_input, row["DMS_filename"]))
            f["offset"] = int(row["start_idx"]) if "start_idx" in mapping.columns and row["start_idx"] != "" else int(args.offset_idx)
            f["mutants"] = [str(m) for m in f["df"][mutant_col]]
            f["positions"] = np.arange(len(seq) + 2, dtype=np.int32) if args.all_positions else \
                pesm.positions_read(f["mutants"], seq, f["offset"])
        except BaseException as e:                  # noqa: BLE001
            if isinstance(e, KeyboardInterrupt):
                raise
            failed[a] = _describe(e)
        frames.append(f)
    failed, = _exchange_reports(world, failed)
    for a in failed:
        frames[a]["positions"] = np.zeros(0, np.int32)
        if rank == 0:
            print(f"assay {frames[a]['dms_id']} FAILED: {failed[a]}", flush=True)
    items, assignment = pdist.plan_position_chunks([len(f["seq"]) for f in frames], [f["positions"] for f in frames],
                                                   world, chunk_forwards=args.chunk_forwards)
    n_toks = [len(f["seq"]) + 2 for f in frames]
    dev = None
    if world > 1:
        import torch.distributed as tdist
        dev = "cuda" if tdist.get_backend() == "nccl" else "cpu"

    def fill(model, local, which, report_overflow):
        """This rank's chunks of the assays in ``which`` -> rows of ``local``; returns ({assay: message}, {assays that overflowed})."""
        bad, over = {}, set()
        for k in assignment[rank]:
            a, chunk = items[k]
            if a not in which or a in bad or a in over:
                continue
            try:
                local[a][chunk] = model.table_rows(frames[a]["seq"], chunk, frames[a]["offset"])
            except BaseException as e:              # noqa: BLE001
                if isinstance(e, KeyboardInterrupt):
                    raise
                if report_overflow and is_overflow(e):
                    over.add(a)
                else:
                    bad[a] = _describe(e)
                    print(f"[rank {rank}] assay {frames[a]['dms_id']} FAILED: {bad[a]}", flush=True)
        return bad, over

    for ci, loc in enumerate(args.model_location):
        live = {a for a in range(len(frames)) if a not in failed}
        local = {a: np.full((n_toks[a], 33), np.nan, dtype=np.float32) for a in range(len(frames))}
        model, bad, over = None, {}, set()
        try:
            model = _seam(make_model, loc) if make_model is not None else _DeviceTables(loc, local_rank, args.precision)
            bad, over = fill(model, local, live, args.precision != "fp32")
        except BaseException as e:                  # noqa: BLE001 -- a checkpoint this rank cannot load fails the assays it holds chunks of
            if isinstance(e, KeyboardInterrupt):
                raise
            bad = {items[k][0]: f"checkpoint {loc}: {_describe(e)}" for k in assignment[rank] if items[k][0] in live}
        if model is not None and hasattr(model, "close"):
            model.close()
        bad, over = _exchange_reports(world, bad, {a: 1 for a in over})
        failed.update({a: m for a, m in bad.items() if a not in failed})
        redo = {a for a in over if a not in failed}
        if redo:                                    # the same set on every rank
            if rank == 0:
                print(f"{cols[ci]}: {args.precision} left the fp16 range in {sorted(frames[a]['dms_id'] for a in redo)}: "
                      "those assays are re-scored in fp32 on every rank", flush=True)
            for a in redo:
                local[a][:] = np.nan
                used[(a, ci)] = "fp32"
            bad = {}
            try:
                model32 = _seam(make_model, loc, "fp32") if make_model is not None else _DeviceTables(loc, local_rank, "fp32")
                if model32 is None:
                    raise pesm.PgmiError("no fp32 model available for the retry")
                bad, _ = fill(model32, local, redo, False)
                if hasattr(model32, "close"):
                    model32.close()
            except BaseException as e:              # noqa: BLE001
                if isinstance(e, KeyboardInterrupt):
                    raise
                bad = {items[k][0]: f"fp32 retry of {loc}: {_describe(e)}" for k in assignment[rank] if items[k][0] in redo}
            bad, = _exchange_reports(world, bad)
            failed.update({a: m for a, m in bad.items() if a not in failed})
        tables = pdist.gather_tables(local, n_toks, device=dev)
        if rank == 0:
            for a, f in enumerate(frames):
                if a in failed:
                    continue
                try:
                    f["df"][cols[ci]] = pesm.score_from_table(tables[a], f["mutants"], f["seq"], f["offset"])
                except BaseException as e:          # noqa: BLE001
                    if isinstance(e, KeyboardInterrupt):
                        raise
                    failed[a] = _describe(e)
    failed, = _exchange_reports(world, failed if rank == 0 else {})            # rank 0's view decides the exit code everywhere
    if rank == 0:
        n_mut = 0
        vectors = []
        for a, f in enumerate(frames):
            df = f["df"]
            if a in failed:
                vectors.append(np.zeros((len(cols), 0 if df is None else len(df))))
                continue
            if ens_cols:
                df["Ensemble_ESM1v"] = sum(df[c] for c in cols) / len(cols)
            out = os.path.join(args.dms_output, f["dms_id"] + ".csv")
            df.to_csv(out + ".tmp", index=False)
            os.replace(out + ".tmp", out)
            n_mut += len(df)
            vectors.append(np.stack([df[c].to_numpy(dtype=np.float64) for c in cols]))
        by_index = {todo[a]: m for a, m in failed.items()}
        if world > 1 or failed or used:
            _write_csv(pd.DataFrame(_summary_rows(mapping, todo, cols, vectors, by_index,
                                                  {(todo[a], c): p for (a, c), p in used.items()})),
                       os.path.join(args.dms_output, "scores_summary.csv"))
        dt = time.time() - t0
        print(f"scored {len(frames) - len(failed)} assays / {n_mut} mutants x {len(cols)} checkpoint(s) on {world} GPU(s), "
              f"{len(items)} position chunks, in {dt:.1f}s = {n_mut / max(dt, 1e-9):.1f} mutants/s (ensemble rate)")
    if world > 1:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()
    _exit_on_failures(mapping, {todo[a]: m for a, m in failed.items()}, rank)


class _DeviceTables:
    """table_rows() on the GPU: an Assay restricted to the chunk's positions (the alignment of work items to
    masked positions is the device path's own: windows, chunking and the head run as in pgmi_assay_run)."""

    def __init__(self, location, device, precision):
        self.model, self.alphabet = pesm.load_model_and_alphabet(location, device=device, precision=precision)

    def table_rows(self, seq, positions, offset):
        first = seq[0] + str(offset) + ("A" if seq[0] != "A" else "C")          # any valid mutant: only the table is read
        a = pesm.Assay(self.model, seq, [first], offset_idx=offset, alphabet=self.alphabet, positions=positions)
        try:
            _, table = a.run(want_table=True)
        finally:
            a.close()
        return table[np.asarray(positions)]

    def close(self):
        self.model.close()


if __name__ == "__main__":
    main(create_parser().parse_args())
