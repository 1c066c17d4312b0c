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
WITHOUT its assay index.  One process per GPU.

Tranception (default ``--shard mutants``; BASELINE config 4, SURVEY 8e "Tranception and pseudo-ppl shard the same way by
mutant chunk"): the unit of work is a CHUNK OF MUTANT ROWS of one assay, priced by the tokens it forwards -- one assay of
the substitution benchmark holds 26 % of the cost (SPG1_STRSG_Olson_2014, 536 962 rows), whole assays cannot balance eight
GPUs.  Chunks are LPT-balanced over the ranks, every rank loads the checkpoint ONCE and keeps it resident across its chunks
(the per-assay retrieval prior is swapped in), scores leave through ONE fixed-stride all_gather (three float64 per row; RCCL
over xGMI), and the rank that scored most rows of an assay assembles and writes its CSV.  Every sequence is scored exactly as
the single-assay CLI scores it (a row's bits do not depend on what shares its batch), so the CSVs are identical to the
unsharded ones.  The reference loop being sharded: tranception/utils/scoring_utils.py:77-150 (a DataLoader over mutated
sequences, independent per sequence).

``--shard assay`` (the MSA Transformer default): assays are independent units, LPT-balanced over the ranks by an
algorithmic cost estimate, every rank runs the single-assay CLI for its own assays and writes their CSVs: no data-path
collective, one reduction at the end.  The N > 1 data path needs torch.distributed (backend nccl = RCCL on ROCm).
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


# ---- Tranception: chunks of mutant rows -------------------------------------------------------------------------------
CHUNKS_PER_RANK = 16        # planning granularity: no chunk costs more than 1 / (16 world) of the whole job
MIN_CHUNK_ROWS = 256        # below this the wild-type windows scored with every chunk stop being negligible


def chunk_cost(T: int, rows: int) -> float:
    """Tokens a chunk forwards (both directions cancel in the ratio), attention share included, plus the wild-type sequence(s)
    scored with it (one per distinct window: ~1 for proteins inside the context)."""
    return (rows + 1.0) * T * (1.0 + T / 7680.0)


def plan_mutant_chunks(seq_lens, n_rows, world: int, max_chunk_rows: int = 0):
    """Work list of (assay position k in the input lists, row0, row1) + its LPT assignment.  Deterministic: every rank
    computes the same plan from the reference table alone.  Small assays stay whole (one checkpoint-resident rank, one
    retrieval prior); large ones are cut so that no item exceeds 1 / (CHUNKS_PER_RANK * world) of the total cost."""
    Ts = [min(int(L) + 2, 1024) for L in seq_lens]
    total = sum(chunk_cost(T, n) for T, n in zip(Ts, n_rows))
    target = total / max(1, CHUNKS_PER_RANK * world)
    items, costs = [], []
    for k, (T, n) in enumerate(zip(Ts, n_rows)):
        n = int(n)
        pieces = 1 if world == 1 else max(1, min(-(-n // MIN_CHUNK_ROWS), int(-(-chunk_cost(T, n) // max(target, 1e-9)))))
        if max_chunk_rows > 0:                                # an explicit cap (tests; a user bounding host memory per call)
            pieces = max(pieces, -(-n // max_chunk_rows))
        for c in range(pieces):
            r0, r1 = n * c // pieces, n * (c + 1) // pieces
            if r1 > r0 or n == 0:
                items.append((k, r0, r1))
                costs.append(chunk_cost(T, r1 - r0))
    return items, pdist.lpt_partition(costs, world), costs


def rows_per_assay(mapping, indices, data_folder):
    """Row count of every assay FILE as the scorer's own parser sees it (``pd.read_csv``: quoted fields with embedded newlines,
    whitespace-only lines and footers count the way ``_assay_frame`` will count them), -1 for a file that cannot be read or
    parsed.  The numbers shape the chunk plan and the all_gather, so they must be the same on every rank: rank 0 counts and
    ``planned_rows`` broadcasts.  The reference table's DMS_total_number_mutants stands in for an unreadable file only so
    that the plan has a cost for it; the rank that then loads it fails that assay (a file nobody can read is a failed assay,
    not an empty one)."""
    out = []
    for i in indices:
        row = mapping.iloc[i]
        try:
            out.append(len(pd.read_csv(os.path.join(data_folder, str(row["DMS_filename"])), usecols=[0], low_memory=False)))
        except Exception:
            out.append(-1)
    return out


def planned_rows(mapping, indices, data_folder, rank, world):
    """(rows per assay for the plan, readable flags): counted once, by rank 0, and broadcast -- every rank plans from the same
    numbers even if the folder looks different from another rank's mount."""
    counts = rows_per_assay(mapping, indices, data_folder) if rank == 0 else None
    if world > 1:
        import torch.distributed as tdist
        box = [counts]
        tdist.broadcast_object_list(box, src=0)
        counts = box[0]
    readable = [c >= 0 for c in counts]
    table = mapping["DMS_total_number_mutants"] if "DMS_total_number_mutants" in mapping.columns else None
    rows = []
    for i, c in zip(indices, counts):
        if c < 0:
            v = table.iloc[i] if table is not None else float("nan")
            c = int(v) if v == v else 0
        rows.append(c)
    return rows, readable


def _close_retrieval(state):
    """Takes a retrieval state's aligner (indel scoring with retrieval: three files per assay, a temporary folder when the alignment
    folder is read-only) away explicitly when the state is replaced."""
    aligner = state.get("aligner") if isinstance(state, dict) else None
    if aligner is not None and hasattr(aligner, "close"):
        aligner.close()


def _lock_host() -> str:
    """Host identity for the lock files: node name + the pid namespace (two containers on one node do not share pids)."""
    import socket
    try:
        ns = os.readlink("/proc/self/ns/pid")
    except OSError:
        ns = ""
    return f"{socket.gethostname()}/{ns}"


def shared_retrieval(ptr, retrieval_args, cache_dir, tag, wait_s=1800.0):
    """``tranception.build_retrieval`` computed ONCE per assay across the ranks of a job: the chunks of a large assay land on
    several ranks, each of which would otherwise re-read the alignment (and recompute sequence weights when no weight file
    exists).  The rank that creates ``<tag>.lock`` first builds the log-prior and publishes it as ``<tag>.npy`` (atomic
    rename); the others wait for the file.  The array is the builder's own output, so every rank scores with the same bits.
    A builder that FAILS (missing alignment or weight file, ragged a2m ...) publishes ``<tag>.failed`` with the message: the
    failure is deterministic, so the waiters raise it at once instead of polling for ``wait_s`` and rebuilding into the same
    error.  A waiter whose builder vanished without either file (killed: the lock's pid is gone, or ``wait_s`` is over) builds
    the prior itself."""
    import numpy as np
    if not retrieval_args:
        return None
    if retrieval_args.get("retrieval_aggregation_mode") == "aggregate_indel":
        return ptr.build_retrieval(retrieval_args)        # carries a per-rank aligner (files of its own under <alignment folder>/Sampled)
    os.makedirs(cache_dir, exist_ok=True)
    path, lock, failed = (os.path.join(cache_dir, tag + ext) for ext in (".npy", ".lock", ".failed"))

    def from_file():
        return dict(log_prior=np.load(path), MSA_start=int(retrieval_args["MSA_start"]), MSA_end=int(retrieval_args["MSA_end"]),
                    weight=float(retrieval_args.get("retrieval_inference_weight", 0.6)))

    def builder_alive():
        """The lock holds 'hostname:pid'.  A pid can only be probed from its own host (and pid namespace): a waiter on another
        node -- or in another container -- sharing the output folder cannot tell, so it keeps polling until the prior, the
        .failed note or ``wait_s`` arrives instead of declaring the builder dead and rebuilding beside it."""
        try:
            host, _, pid = open(lock).read().strip().rpartition(":")
            pid = int(pid or 0)
        except (OSError, ValueError):
            return True                                   # lock just created, not written yet (or unreadable): keep waiting
        if pid <= 0 or host != _lock_host():
            return True
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        except OSError:
            return True
        return True
    if os.path.exists(path):
        return from_file()
    try:
        fd = os.open(lock, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
        os.write(fd, f"{_lock_host()}:{os.getpid()}".encode())
        os.close(fd)
        builder = True
    except FileExistsError:
        builder = False
    if not builder:
        t_end = time.time() + wait_s
        while time.time() < t_end:
            if os.path.exists(path):
                return from_file()
            if os.path.exists(failed):
                raise RuntimeError(f"retrieval prior of {tag}: the building rank failed: {open(failed).read().strip()}")
            if not builder_alive():
                break
            time.sleep(0.05)
    try:
        r = ptr.build_retrieval(retrieval_args)
    except BaseException as e:
        if not isinstance(e, KeyboardInterrupt):
            tmp = f"{failed}.{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                f.write(f"{type(e).__name__}: {e}")
            os.replace(tmp, failed)
        raise
    tmp = f"{path}.{os.getpid()}.tmp.npy"
    np.save(tmp, r["log_prior"])
    os.replace(tmp, path)
    return r


def chunk_row_scores(model, chunk, wild_type, scoring_mirror, indel_mode):
    """[rows, 3] float64 (avg_score_L_to_R, avg_score_R_to_L, avg_score) of a chunk of assay rows, NaN for rows that are
    the wild type itself (the scorer drops them; the CSV gets its zero row from ``assemble_scores``).  ``chunk`` holds
    mutated_sequence (+ mutant)."""
    import numpy as np
    res = model.score_mutants(DMS_data=chunk, target_seq=wild_type, scoring_mirror=scoring_mirror, indel_mode=indel_mode,
                              append_wildtype_row=False)
    res = res.drop_duplicates("mutated_sequence").set_index("mutated_sequence")
    cols = ["avg_score_L_to_R", "avg_score_R_to_L" if scoring_mirror else "avg_score_L_to_R", "avg_score"]
    out = np.full((len(chunk), 3), np.nan, dtype=np.float64)
    seqs = chunk["mutated_sequence"].to_numpy()
    hit = np.array([sq in res.index for sq in seqs], dtype=bool)
    if hit.any():
        out[hit] = res.loc[seqs[hit], cols].to_numpy(dtype=np.float64)
    return out


def assemble_scores(frame, per_row, wild_type, scoring_window, scoring_mirror, indel_mode):
    """The frame ``TranceptionModel.score_mutants`` returns for the whole assay, rebuilt from per-row scores: one row per
    distinct non-wild-type sequence in first-appearance order ('sliding': sorted, the reference's groupby order), then the
    wild type's zero row when the input lists it (model_pytorch.py:915-924)."""
    import numpy as np
    seqs = frame["mutated_sequence"].to_numpy()
    keep = seqs != wild_type
    out = pd.DataFrame({"mutated_sequence": seqs[keep], "avg_score_L_to_R": per_row[keep, 0]})
    if scoring_mirror:
        out["avg_score_R_to_L"] = per_row[keep, 1]
    out["avg_score"] = per_row[keep, 2]
    out = out.drop_duplicates("mutated_sequence", keep="first")
    if scoring_window == "sliding":
        out = out.sort_values("mutated_sequence", kind="stable")
    out = out.reset_index(drop=True)
    key = "mutant" if indel_mode else "mutated_sequence"
    if wild_type in frame[key].values:
        names = [key, "avg_score_L_to_R"] + (["avg_score_R_to_L"] if scoring_mirror else []) + ["avg_score"]
        out = pd.concat([out, pd.DataFrame([[wild_type] + [0] * (len(names) - 1)], columns=names)], ignore_index=True)
    return out


def _assay_frame(args, assay_file, wild_type):
    """The assay file as score_mutants prepares it (model_pytorch.py:888-894): mutated_sequence (+ mutant) per row."""
    from . import tranception as ptr
    df = pd.read_csv(args.DMS_data_folder + os.sep + assay_file, low_memory=False)
    key = "mutant" if args.indel_mode else "mutated_sequence"
    if key not in df:               # the single-assay scorer ends in DMS_data[key] (model_pytorch.py:915-919): same failure, before the work
        raise KeyError(key)
    if "mutated_sequence" not in df and not args.indel_mode:
        df["mutated_sequence"] = ptr.mutated_sequences(wild_type, df["mutant"])
    assert ("mutated_sequence" in df), "DMS file to score does not have mutated_sequence column"
    if "mutant" not in df:
        df["mutant"] = df["mutated_sequence"]
    return df[["mutated_sequence", "mutant"]]


def main_tranception_mutants(own, rest, mapping, indices, rank, local_rank, world, make_model=None):
    """``make_model`` is a test seam: (checkpoint, device, scoring_window) -> object with a settable ``retrieval``,
    score_mutants(...) and close()."""
    import numpy as np
    from . import score_tranception_proteingym as cli
    from . import tranception as ptr
    t0 = time.time()
    base = cli.create_parser().parse_args(rest + ["--device", str(local_rank)])
    if base.model_framework != "pytorch":
        raise NotImplementedError("only --model_framework pytorch has an MI355X backend")
    n_rows, readable = planned_rows(mapping, indices, base.DMS_data_folder, rank, world)
    items, assignment, costs = plan_mutant_chunks([len(str(mapping.iloc[i]["target_seq"])) for i in indices], n_rows, world,
                                                  max_chunk_rows=own.max_chunk_rows)
    mine = assignment[rank]
    loads = [sum(costs[k] for k in part) for part in assignment]
    print(f"[rank {rank}/{world}] tranception: {len(mine)} of {len(items)} mutant chunks ({len({items[k][0] for k in mine})} assays), "
          f"planned load max/mean {max(loads) / max(sum(loads) / world, 1e-30):.4f}", flush=True)
    if own.dry_run:
        if world > 1:
            import torch.distributed as tdist
            tdist.barrier()
            tdist.destroy_process_group()
        return [items[k] for k in mine]
    mirror = not base.deactivate_scoring_mirror
    model = None
    local, failed, frames = {}, {}, {}
    # one tag per job so that a later run (other weights, other alignment) never reads this one's priors: rank 0's start time
    job_tag = str(int(t0 * 1000))
    if world > 1:
        import torch
        import torch.distributed as tdist
        stamp = torch.tensor([int(t0 * 1000)], dtype=torch.int64, device="cuda" if tdist.get_backend() == "nccl" else "cpu")
        tdist.broadcast(stamp, 0)
        job_tag = str(int(stamp.item()))

    def assay_inputs(k):
        args = cli.create_parser().parse_args(rest + ["--DMS_index", str(indices[k]), "--device", str(local_rank)])
        dms_id, wild_type, assay_file, msa = cli.resolve_inputs(args)
        return args, str(dms_id), wild_type, assay_file, msa

    by_assay = {}
    for j in mine:
        by_assay.setdefault(items[j][0], []).append(j)
    for k, js in by_assay.items():
        try:
            args, dms_id, wild_type, assay_file, msa = assay_inputs(k)
            if not readable[k]:
                raise OSError(f"{assay_file}: the assay file could not be read when the job was planned")
            frame = _assay_frame(args, assay_file, wild_type)
            if len(frame) != n_rows[k]:
                raise ValueError(f"{assay_file}: {len(frame)} rows now, {n_rows[k]} when the job was planned (rank 0's count)")
            frames[k] = frame
            if n_rows[k] == 0:                               # a header-only file: nothing to score, the CSV below is header-only too
                for j in js:
                    local[j] = np.zeros(0)
                continue
            if model is None:                                # one checkpoint load per rank
                model = make_model(base.checkpoint, local_rank, base.scoring_window) if make_model is not None else \
                    ptr.from_pretrained(base.checkpoint, device=local_rank, scoring_window=base.scoring_window)
            _close_retrieval(getattr(model, "retrieval", None))  # the previous assay's aligner files, now (not when __del__ gets to it)
            if world > 1:                                    # an assay's chunks may sit on several ranks: one of them builds the prior
                model.retrieval = shared_retrieval(ptr, cli.retrieval_arguments(args, wild_type, msa),
                                                   os.path.join(args.output_scores_folder, ".retrieval_prior_cache"), f"{job_tag}_{dms_id}")
            else:
                model.retrieval = ptr.build_retrieval(cli.retrieval_arguments(args, wild_type, msa))
            for j in js:
                _, r0, r1 = items[j]
                local[j] = chunk_row_scores(model, frame.iloc[r0:r1], wild_type, mirror, args.indel_mode).ravel()
        except BaseException as e:                           # keep the collective below in step on every rank
            if isinstance(e, KeyboardInterrupt):
                raise
            failed[k] = f"{type(e).__name__}: {e}"
            print(f"[rank {rank}] assay {indices[k]} FAILED: {failed[k]}", flush=True)
            for j in js:
                local[j] = np.full(3 * (items[j][2] - items[j][1]), np.nan)
    if model is not None:
        _close_retrieval(getattr(model, "retrieval", None))
        model.close()
    dev = None
    bad = np.zeros(len(indices), dtype=np.int64)
    for k in failed:
        bad[k] = 1
    if world > 1:
        import torch
        import torch.distributed as tdist
        dev = "cuda" if tdist.get_backend() == "nccl" else "cpu"
        flag = torch.from_numpy(bad).to(dev)
        tdist.all_reduce(flag)                               # which assays lost a chunk anywhere
        bad = flag.cpu().numpy()
    allv = pdist.gather_score_vectors(local, [3 * (r1 - r0) for _, r0, r1 in items], assignment, device=dev)
    # the rank that scored most rows of an assay writes it (ties: the lower rank); it already holds the assay's frame
    rows_by = np.zeros((len(indices), world), dtype=np.int64)
    for r, part in enumerate(assignment):
        for j in part:
            rows_by[items[j][0], r] += items[j][2] - items[j][1] + 1
    written = 0
    for k in range(len(indices)):
        if int(np.argmax(rows_by[k])) != rank or bad[k]:
            continue
        args, dms_id, wild_type, assay_file, _ = assay_inputs(k)
        frame = frames[k] if k in frames else _assay_frame(args, assay_file, wild_type)
        per_row = np.concatenate([allv[j] for j, it in enumerate(items) if it[0] == k] or [np.zeros(0)]).reshape(-1, 3)
        out = assemble_scores(frame, per_row, wild_type, base.scoring_window, mirror, args.indel_mode)
        os.makedirs(args.output_scores_folder, exist_ok=True)
        out_csv = args.output_scores_folder + os.sep + dms_id + ".csv"
        out.to_csv(out_csv + ".tmp", index=False)
        os.replace(out_csv + ".tmp", out_csv)
        written += 1
    print(f"[rank {rank}] {sum(items[j][2] - items[j][1] for j in mine)} rows scored, {written} CSVs written in {time.time() - t0:.1f}s", flush=True)
    n_failed = int((bad > 0).sum())
    if world > 1:
        import torch.distributed as tdist
        tdist.barrier()
        if rank == 0:                                        # the job's shared priors have served every rank
            cache = os.path.join(base.output_scores_folder, ".retrieval_prior_cache")
            if os.path.isdir(cache):
                for fn in os.listdir(cache):
                    if fn.startswith(job_tag + "_"):
                        os.remove(os.path.join(cache, fn))
                if not os.listdir(cache):
                    os.rmdir(cache)
        tdist.destroy_process_group()
    if n_failed:
        raise SystemExit(f"run_sharded: {n_failed} assay(s) failed (see the per-rank messages); their CSVs were not written")
    return [items[k] for k in mine]


def _value_after(argv, flag):
    for i, a in enumerate(argv):
        if a == flag and i + 1 < len(argv):
            return argv[i + 1]
    raise SystemExit(f"run_sharded: the single-assay arguments must contain {flag}")


OUTPUT_FLAG = {"tranception": ("--output_scores_folder", "avg_score"), "msa_transformer": ("--dms-output", None)}


def _without_existing(baseline, rest, mapping, indices, rank, world):
    """--skip-existing: the assays still to do, decided by rank 0's view of the output folder and broadcast (the plan and every
    collective are shaped by this list).  Tranception: a CSV with an ``avg_score`` column; MSA Transformer: a CSV that holds the
    ``<checkpoint>_ensemble`` column its CLI writes after the last seed (seeds already on disk are skipped by the CLI itself)."""
    flag, column = OUTPUT_FLAG[baseline]
    todo = []
    if rank == 0:
        folder = _value_after(rest, flag)
        for i in indices:
            path = os.path.join(folder, str(mapping.iloc[i]["DMS_id"]) + ".csv")
            done = False
            if os.path.exists(path):
                try:
                    have = list(pd.read_csv(path, nrows=0).columns)
                    done = (column in have) if column else any(str(c).endswith("_ensemble") for c in have)
                except Exception:                      # noqa: BLE001 -- a torn file is not a finished assay
                    done = False
            if not done:
                todo.append(i)
        print(f"run_sharded: --skip-existing leaves {len(todo)} of {len(indices)} assays to do", flush=True)
    if world > 1:
        import torch.distributed as tdist
        box = [todo]
        tdist.broadcast_object_list(box, src=0)
        todo = box[0]
    return todo


def plan(baseline: str, mapping: pd.DataFrame, indices, world: int):
    costs = [assay_cost(baseline, mapping.iloc[i]) for i in indices]
    assignment = pdist.lpt_partition(costs, world)
    return [[indices[k] for k in part] for part in assignment]


def main(argv=None, make_model=None):
    """``make_model``: test seam of --shard mutants (see main_tranception_mutants)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    if "--" not in argv:
        raise SystemExit(__doc__)
    cut = argv.index("--")
    ap = argparse.ArgumentParser()
    ap.add_argument("baseline", choices=sorted(BASELINES))
    ap.add_argument("--indices", type=int, nargs="*", default=None, help="default: every row of the reference file")
    ap.add_argument("--dry-run", action="store_true", help="print this rank's assays and exit")
    ap.add_argument("--shard", choices=["assay", "positions", "mutants"], default=None,
                    help="unit of work.  tranception: 'mutants' (default) = chunks of an assay's rows, resident model, one all_gather "
                         "of the scores; 'assay' = whole assays through the single-assay CLI.  msa_transformer: 'assay' (default), or "
                         "'positions' = every rank works on EVERY assay and forwards its share of the (seed, masked position) pairs; "
                         "the log-prob tables are all_gathered (one BLAT-size assay x 5 seeds is ~190 s of serial work: sharding whole "
                         "assays cannot balance a handful of them)")
    ap.add_argument("--max-chunk-rows", type=int, default=0, help="--shard mutants: cap on the rows of one work item (default: planned from the cost)")
    ap.add_argument("--backend", type=str, default=None)
    ap.add_argument("--skip-existing", action="store_true",
                    help="leave out the assays whose <output folder>/<DMS_id>.csv already exists with a score column (a re-run after a "
                         "job that died or failed in some assays takes up the missing ones; decided on rank 0, the same list on every rank)")
    own = ap.parse_args(argv[:cut])
    rest = argv[cut + 1:]
    module, index_flag, ref_flag, device_flag = BASELINES[own.baseline]
    if index_flag in rest:
        raise SystemExit(f"run_sharded: do not pass {index_flag}; assays are distributed over the ranks")
    if own.shard is None:
        own.shard = "mutants" if own.baseline == "tranception" else "assay"
    if own.shard == "positions" and own.baseline != "msa_transformer":
        raise SystemExit("run_sharded: --shard positions is implemented for msa_transformer")
    if own.shard == "mutants" and own.baseline != "tranception":
        raise SystemExit("run_sharded: --shard mutants is implemented for tranception")
    rank, local_rank, world = pdist.init_from_env(own.backend)
    mapping = pd.read_csv(_value_after(rest, ref_flag))
    indices = list(range(len(mapping))) if own.indices is None else list(own.indices)
    if own.skip_existing:
        indices = _without_existing(own.baseline, rest, mapping, indices, rank, world)
    if own.shard == "mutants":
        return main_tranception_mutants(own, rest, mapping, indices, rank, local_rank, world, make_model=make_model)
    by_position = own.shard == "positions"
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
                # --shard positions: every rank runs collectives INSIDE mod.main for every assay; a failure on this rank alone
                # (an out-of-memory, a device error) would leave the peers waiting in an all_gather or pair this rank's next
                # collective with their current one.  Fail the whole job instead: the launcher tears the peers down.
                if isinstance(e, KeyboardInterrupt) or (by_position and world > 1):
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
