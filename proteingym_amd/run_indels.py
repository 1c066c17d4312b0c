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
batch) and ONE fixed-stride all_gather (RCCL over xGMI; 8 bytes per mutant) returns the score vector to every rank;
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
from .run_benchmark import column_names, _finish_frame, _write_csv


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


def main(args, make_model=None):
    """``make_model`` is a test seam: (location) -> object with score(sequences) -> float64 array, and close()."""
    rank, local_rank, world = pdist.init_from_env(args.backend)
    mapping = pd.read_csv(args.dms_mapping)
    indices = list(range(len(mapping))) if args.dms_indices is None else list(args.dms_indices)
    cols, ens_cols = column_names(args.model_location, args.model_type)
    os.makedirs(args.dms_output, exist_ok=True)
    t0 = time.time()
    frames, pool, spans = [], [], []
    for i in indices:                                           # every rank reads the (small) input files: the pool
        row = mapping.iloc[i]                                   # and its partition must be identical everywhere
        df = pd.read_csv(os.path.join(args.dms_input, row["DMS_filename"]))
        if args.sequence_col not in df:
            raise ValueError(f"{row['DMS_filename']}: no '{args.sequence_col}' column (indel assays carry the full mutated sequence)")
        frames.append((str(row["DMS_id"]), df))
        spans.append((len(pool), len(pool) + len(df)))
        pool.extend(str(s) for s in df[args.sequence_col])
    lengths = [len(s) for s in pool]
    assignment, loads = partition_pool(lengths, world)
    mine = assignment[rank]
    dev = None
    if world > 1:
        import torch.distributed as tdist
        dev = "cuda" if tdist.get_backend() == "nccl" else "cpu"
    vectors = []
    for loc in args.model_location:
        model = make_model(loc) if make_model is not None else _DevicePppl(loc, local_rank, args.precision)
        local = np.asarray(model.score([pool[k] for k in mine]), dtype=np.float64)
        model.close()
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
        for (dms_id, df), (a, b) in zip(frames, spans):
            _write_csv(_finish_frame(df, cols, ens_cols, [v[a:b] for v in vectors]), os.path.join(args.dms_output, dms_id + ".csv"))
        dt = time.time() - t0
        rows = sum(max(0, L - 2) for L in lengths)
        print(f"pseudo-ppl: {len(frames)} assays / {len(pool)} sequences / {rows} masked forwards x {len(cols)} checkpoint(s) on "
              f"{world} GPU(s) in {dt:.1f}s = {len(pool) / max(dt, 1e-9):.1f} mutants/s; planned load max/mean "
              f"{loads.max() / max(loads.mean(), 1e-30):.4f}")
    if world > 1:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()
    return vectors


if __name__ == "__main__":
    main(create_parser().parse_args())
