"""Pseudo-perplexity scoring of variable-length (indel) mutant libraries on all GPUs of a node -- BASELINE config 5.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m proteingym_amd.run_indels --model-location esm2_t33_650M_UR50D.pt --model_type ESM2 \\
        --dms_mapping reference_files/DMS_indels.csv --dms-input DMS_ProteinGym_indels \\
        --dms-output scores/ESM2_indels [--dms_indices 0 1 2 ...]

What it replaces: one ``compute_fitness.py --scoring-strategy pseudo-ppl --dms_index i`` process per assay
(/root/reference/proteingym/baselines/esm/compute_fitness.py:515-529, compute_pppl :258-279), i.e. one batch-1
forward per (mutant, residue).  Cost is dominated by ONE assay (CAPSD_AAV2S designed: 225 998 of the 287 207 indel
mutants, 735 residues), so assays are not the unit of work here: the ``mutated_sequence`` rows of ALL selected assays
form one pool, every sequence is priced by its algorithmic FLOPs ((L-2) forwards of L+2 tokens), the pool is
LPT-balanced over the ranks (one process per GPU, weights replicated), each rank uploads only its share as a
device-resident library (esm.SequenceLibrary -> pgmi_pppl_*: rows enumerated on the device, mixed lengths packed per
batch), slice by slice with the scores so far saved after each (--save-every-forwards: a re-run takes up where it ended), and ONE
fixed-stride all_gather (RCCL over xGMI; 8 bytes per mutant) returns the score vector to every rank;
rank 0 writes one ``<DMS_id>.csv`` per assay with the reference's columns.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import pandas as pd

from . import dist as pdist
from . import esm as pesm
from .run_benchmark import (column_names, _finish_frame, _write_csv, _describe, _exchange_reports, is_overflow, _seam,
                            _summary_rows, _exit_on_failures)


def create_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--model-location", type=str, nargs="+", required=True)
    p.add_argument("--model_type", type=str, nargs="+", default=["ESM2"])
    p.add_argument("--dms_mapping", type=str, required=True)
    p.add_argument("--dms-input", type=str, required=True)
    p.add_argument("--dms-output", type=str, required=True)
    p.add_argument("--dms_indices", type=int, nargs="*", default=None, help="default: every row of the mapping")
    p.add_argument("--sequence-col", type=str, default="mutated_sequence")
    p.add_argument("--precision", type=str, default="f16x3", choices=sorted(pesm._lib.PRECISIONS))
    p.add_argument("--backend", type=str, default=None, help="torch.distributed backend (default nccl)")
    p.add_argument("--save-every-forwards", type=int, default=250000,
                   help="a rank scores its share in slices of about this many masked forwards (~10 min of one GPU at 735 residues) and "
                        "saves the scores it has after each slice under <dms-output>/.partial; a re-run of the same command (same files, "
                        "checkpoints, precision, number of ranks) takes up where the slices ended -- the table is days of GPU time, a job "
                        "that dies at hour 17 must not start over.  A sequence's score does not depend on its slice: same bits.  0 = off")
    return p


def sequence_cost(length: int, **model_dims) -> float:
    """Algorithmic FLOPs of compute_pppl for one sequence: range(1, L-1) masked forwards of L+2 tokens."""
    return max(0, length - 2) * pdist.forward_flops(length + 2, **model_dims)


def partition_pool(lengths, world: int, **model_dims):
    """LPT over the pooled sequences by algorithmic cost.  Deterministic: every rank computes the same
    {rank -> sorted pool indices}.  Returns (assignment, planned load per rank)."""
    cost = {}
    costs = [cost.setdefault(int(L), sequence_cost(int(L), **model_dims)) for L in lengths]
    assignment = pdist.lpt_partition(costs, world)
    loads = np.array([sum(costs[k] for k in part) for part in assignment], dtype=np.float64)
    return assignment, loads


class _DevicePppl:
    """One checkpoint on this rank's GPU: score(sequences) = SequenceLibrary.score()."""

    def __init__(self, location, device, precision):
        self.model, self.alphabet = pesm.load_model_and_alphabet(location, device=device, precision=precision)
        self.stats = None

    def score(self, sequences):
        if not sequences:
            return np.zeros(0)
        lib = pesm.SequenceLibrary(self.model, sequences, self.alphabet)
        try:
            out = lib.score()
            self.stats = lib.stats()
            return out
        finally:
            lib.close()

    def close(self):
        self.model.close()


def share_digest(sequences, location: str, precision: str) -> str:
    """What a saved slice file must match to be taken up: this rank's sequences in order, the checkpoint (name, size) and the precision."""
    import hashlib
    h = hashlib.sha256()
    h.update(("\n".join(sequences)).encode())
    h.update(f"|{os.path.basename(str(location))}|{os.path.getsize(location) if os.path.exists(str(location)) else -1}|{precision}".encode())
    return h.hexdigest()


def score_in_slices(model, sequences, path, forwards_per_save: int, digest: str, log=None):
    """``model.score`` over ``sequences`` in slices of about ``forwards_per_save`` masked forwards; after every slice the scores so far go
    to ``path`` (atomic replace).  A file left there by an earlier run of the same share (``digest``) is taken up: its sequences are not
    scored again.  Returns the float64 scores."""
    n = len(sequences)
    out = np.full(n, np.nan, dtype=np.float64)
    done = 0
    if path and os.path.exists(path):
        try:
            with np.load(path, allow_pickle=False) as z:
                if str(z["digest"]) == digest and z["scores"].shape == (n,):
                    done = int(z["done"])
                    out[:done] = z["scores"][:done]
        except Exception:                                       # noqa: BLE001 -- a torn or foreign file: start over
            done = 0
        if done and log:
            log(f"taking up {done} of {n} sequences from {path}")
    forwards = [max(0, len(s) - 2) for s in sequences]
    while done < n:
        end, acc = done, 0
        while end < n and (end == done or acc + forwards[end] <= forwards_per_save):
            acc += forwards[end]
            end += 1
        out[done:end] = np.asarray(model.score(sequences[done:end]), dtype=np.float64)
        done = end
        if path and done < n:
            tmp = path + ".tmp.npz"
            np.savez(tmp, digest=np.asarray(digest), done=np.asarray(done), scores=out)
            os.replace(tmp, path)
    return out


def main(args, make_model=None):
    """``make_model`` is a test seam: (location) -> object with score(sequences) -> float64 array, and close().

    Failures (run_benchmark, note above ``_Fp32Retry``): an assay whose file cannot be read or lacks the sequence column is
    dropped on every rank before the pool is built; when a rank's share does not score in one piece, it is scored assay by
    assay -- the assay that fails fills NaN; an assay that leaves the fp16 range anywhere is re-scored on fp32 models of the
    same checkpoint on EVERY rank that holds rows of it, so a library never mixes precisions and the scores do not depend on
    the number of ranks (pseudo-ppl is batch-invariant: the other assays' bits do not change) -- and every rank still reaches
    every exchange; failed assays' CSVs are not written and the job exits non-zero."""
    rank, local_rank, world = pdist.init_from_env(args.backend)
    mapping = pd.read_csv(args.dms_mapping)
    indices = list(range(len(mapping))) if args.dms_indices is None else list(args.dms_indices)
    cols, ens_cols = column_names(args.model_location, args.model_type)
    os.makedirs(args.dms_output, exist_ok=True)
    t0 = time.time()
    read, failed, used = {}, {}, {}
    for i in indices:                                           # every rank reads the (small) input files: the pool
        row = mapping.iloc[i]                                   # and its partition must be identical everywhere
        try:
            df = pd.read_csv(os.path.join(args.dms_input, row["DMS_filename"]))
            if args.sequence_col not in df:
                raise ValueError(f"{row['DMS_filename']}: no '{args.sequence_col}' column (indel assays carry the full mutated sequence)")
            read[i] = df
        except BaseException as e:                              # noqa: BLE001
            if isinstance(e, KeyboardInterrupt):
                raise
            failed[i] = _describe(e)
    failed, = _exchange_reports(world, failed)                  # an assay unreadable anywhere is out of the pool everywhere
    frames, pool, spans, owner = [], [], [], []
    for i in indices:
        if i in failed:
            if rank == 0:
                print(f"assay {mapping.iloc[i]['DMS_id']} FAILED: {failed[i]}", flush=True)
            continue
        df = read[i]
        frames.append((i, str(mapping.iloc[i]["DMS_id"]), df))
        spans.append((len(pool), len(pool) + len(df)))
        pool.extend(str(s) for s in df[args.sequence_col])
        owner += [i] * len(df)
    lengths = [len(s) for s in pool]
    assignment, loads = partition_pool(lengths, world)
    mine = assignment[rank]
    dev = None
    if world > 1:
        import torch.distributed as tdist
        dev = "cuda" if tdist.get_backend() == "nccl" else "cpu"

    by_assay = {}
    for j, k in enumerate(mine):
        by_assay.setdefault(owner[k], []).append(j)

    def score_by_assay(model, local, which, report_overflow):
        """This rank's share of the assays in ``which``, one assay per call -> ({assay: message}, {assays that overflowed})."""
        bad, over = {}, set()
        for i in which:
            js = by_assay.get(i, [])
            if not js:
                continue
            try:
                local[js] = np.asarray(model.score([pool[mine[j]] for j in js]), dtype=np.float64)
            except BaseException as e:                          # noqa: BLE001
                if isinstance(e, KeyboardInterrupt):
                    raise
                if report_overflow and is_overflow(e):
                    over.add(i)
                else:
                    bad[i] = _describe(e)
                    print(f"[rank {rank}] assay {mapping.iloc[i]['DMS_id']} x {cols[ci]} FAILED: {bad[i]}", flush=True)
        return bad, over

    vectors = []
    partial_dir, partials = os.path.join(args.dms_output, ".partial"), []
    for ci, loc in enumerate(args.model_location):
        local = np.full(len(mine), np.nan, dtype=np.float64)
        live = [i for i, _, _ in frames if i not in failed]
        bad, over = {}, set()
        model = None
        try:
            model = _seam(make_model, loc) if make_model is not None else _DevicePppl(loc, local_rank, args.precision)
            try:
                if len(live) < len(frames):
                    raise RuntimeError("an assay of the pool failed on an earlier checkpoint")
                share = [pool[k] for k in mine]
                if args.save_every_forwards > 0 and share:
                    os.makedirs(partial_dir, exist_ok=True)
                    partial = os.path.join(partial_dir, f"{cols[ci]}_rank{rank}of{world}.npz")
                    local[:] = score_in_slices(model, share, partial, args.save_every_forwards, share_digest(share, loc, args.precision),
                                               log=lambda m: print(f"[rank {rank}] {cols[ci]}: {m}", flush=True))
                    partials.append(partial)
                else:
                    local[:] = np.asarray(model.score(share), dtype=np.float64)
            except BaseException as e:                          # noqa: BLE001 -- which assay it was shows one at a time
                if isinstance(e, KeyboardInterrupt):
                    raise
                bad, over = score_by_assay(model, local, live, args.precision != "fp32")
        except BaseException as e:                              # noqa: BLE001 -- a checkpoint this rank cannot load
            if isinstance(e, KeyboardInterrupt):
                raise
            bad = {i: f"checkpoint {loc}: {_describe(e)}" for i in by_assay if i in live}
        if model is not None:
            model.close()
        bad, over = _exchange_reports(world, bad, {i: 1 for i in over})
        failed.update({i: m for i, m in bad.items() if i not in failed})
        redo = [i for i in live if i in over and i not in failed]
        if redo:                                                # the same list on every rank: a library never mixes precisions
            if rank == 0:
                print(f"{cols[ci]}: {args.precision} left the fp16 range in {[str(mapping.iloc[i]['DMS_id']) for i in redo]}: "
                      "those assays are re-scored in fp32 on every rank", flush=True)
            bad = {}
            try:
                model32 = _seam(make_model, loc, "fp32") if make_model is not None else _DevicePppl(loc, local_rank, "fp32")
                if model32 is None:
                    raise pesm.PgmiError("no fp32 model available for the retry")
                bad, _ = score_by_assay(model32, local, redo, False)
                model32.close()
            except BaseException as e:                          # noqa: BLE001
                if isinstance(e, KeyboardInterrupt):
                    raise
                bad = {i: f"fp32 retry of {loc}: {_describe(e)}" for i in redo if i in by_assay}
            bad, = _exchange_reports(world, bad)
            failed.update({i: m for i, m in bad.items() if i not in failed})
            used.update({(i, ci): "fp32" for i in redo})
        if world > 1:                                           # one item per rank: its share of the pool
            got = pdist.gather_score_vectors({rank: local}, [len(a) for a in assignment], [[r] for r in range(world)], device=dev)
            full = np.empty(len(pool), dtype=np.float64)
            for r in range(world):
                full[assignment[r]] = got[r]
        else:
            full = np.empty(len(pool), dtype=np.float64)
            full[mine] = local
        vectors.append(full)
    if rank == 0:
        for (i, dms_id, df), (a, b) in zip(frames, spans):
            if i not in failed:
                _write_csv(_finish_frame(df, cols, ens_cols, [v[a:b] for v in vectors]), os.path.join(args.dms_output, dms_id + ".csv"))
        if failed or used:
            span_of = {i: ab for (i, _, _), ab in zip(frames, spans)}
            per = [np.stack([v[slice(*span_of[i])] for v in vectors]) if i in span_of else np.zeros((len(cols), 0)) for i in indices]
            _write_csv(pd.DataFrame(_summary_rows(mapping, indices, cols, per, failed, used)), os.path.join(args.dms_output, "scores_summary.csv"))
        dt = time.time() - t0
        rows = sum(max(0, L - 2) for L in lengths)
        print(f"pseudo-ppl: {len(frames)} assays / {len(pool)} sequences / {rows} masked forwards x {len(cols)} checkpoint(s) on "
              f"{world} GPU(s) in {dt:.1f}s = {len(pool) / max(dt, 1e-9):.1f} mutants/s; planned load max/mean "
              f"{loads.max() / max(loads.mean(), 1e-30):.4f}")
    if world > 1:
        import torch.distributed as tdist
        tdist.barrier()                                         # rank 0 has written the CSVs: the saved slices are no longer needed
        tdist.destroy_process_group()
    if not failed:                                              # (a failed run keeps them: its re-run scores the failed assays again, and
        for f in partials:                                      # takes the shares up only if the pool -- hence the digest -- is unchanged)
            if os.path.exists(f):
                os.remove(f)
        try:
            os.rmdir(partial_dir)
        except OSError:
            pass
    _exit_on_failures(mapping, failed, rank, who="run_indels")
    return vectors


if __name__ == "__main__":
    main(create_parser().parse_args())
