"""Drop-in for ``proteingym/baselines/esm/compute_fitness.py`` (ESM-1v / ESM-1b / ESM2 branch)
running on MI355X through libpgmi.so.

Same flags (including the reference's mixed ``-``/``_`` spellings), same input resolution, same
output file name, same columns, same ensemble rule -- so
``scripts/scoring_DMS_zero_shot/scoring_ESM1v_substitutions.sh`` / ``scoring_ESM2_substitutions.sh``
/ ``scoring_ESM1b_substitutions.sh`` only need their ``python .../compute_fitness.py`` line pointed
here, and ``proteingym/merge.py`` + ``performance_DMS_benchmarks.py`` consume the CSV unchanged.

Reference: /root/reference/proteingym/baselines/esm/compute_fitness.py
  create_parser :100-238   main :282-543   label_row :240-250   compute_pppl :258-279
Additive flags (not in the reference): --device, --precision, --all-positions, --shard-positions.
The MSA-Transformer branch (:358-425) is score_msa_transformer below (masked-marginals, and the pseudo-ppl variant exactly as
the reference behaves -- compute_pppl_msa).
"""
from __future__ import annotations

import argparse
import math
import os
import pathlib
import sys

import numpy as np
import pandas as pd

from . import esm as pesm


# Flag names, types and defaults are the reference's CLI contract (compute_fitness.py:100-238) so that the launchers
# under scripts/scoring_DMS_zero_shot/ keep working; help texts are ours.  Last three entries are additive.
_FLAGS = [
    ("--model_type", dict(type=str, nargs="+", default="MSA_transformer", help="ESM1v | ESM1b | ESM2 | MSA_transformer")),
    ("--model-location", dict(type=str, nargs="+", help="one or more local .pt checkpoints (fair-esm v1/v2 layout)")),
    ("--sequence", dict(type=str, help="unused (kept for compatibility)")),
    ("--dms-input", dict(type=pathlib.Path, help="assay CSV, or with --dms_index the folder holding the assay CSVs")),
    ("--dms_index", dict(type=int, help="row of --dms_mapping to score")),
    ("--dms_mapping", dict(type=str, help="reference CSV (DMS_id, target_seq, DMS_filename, MSA_* ...)")),
    ("--mutation-col", dict(type=str, default="mutant", help="column holding the substitutions, e.g. 'A42G:L77P'")),
    ("--dms-output", dict(type=pathlib.Path, help="output folder; one <DMS_id>.csv per assay")),
    ("--offset-idx", dict(type=int, default=1, help="position of the first residue in the mutation strings")),
    ("--scoring-strategy", dict(type=str, default="wt-marginals", choices=["wt-marginals", "pseudo-ppl", "masked-marginals"],
                                help="how a mutant is scored from the language model")),
    ("--msa-path", dict(type=pathlib.Path, help="alignment file, or with --dms_index the folder of alignments (MSA Transformer)")),
    ("--msa-sampling-strategy", dict(type=str, default="sequence-reweighting", help="sequence-reweighting | random | first_x_rows")),
    ("--msa-samples", dict(type=int, default=400, help="rows sampled from the alignment per seed")),
    ("--msa-weights-folder", dict(type=str, default=None, help="folder of sequence-weight .npy files (sequence-reweighting)")),
    ("--seeds", dict(type=int, nargs="+", default=1, help="one sampled alignment and one score column per seed")),
    ("--filter-msa", dict(action="store_true", help="filter the alignment with hhfilter (--path-to-hhfilter) before sampling")),
    ("--hhfilter-min-cov", dict(type=int, default=75, help="hhfilter -cov")),
    ("--hhfilter-max-seq-id", dict(type=int, default=90, help="hhfilter -id")),
    ("--hhfilter-min-seq-id", dict(type=int, default=0, help="hhfilter -qid")),
    ("--path-to-hhfilter", dict(type=str, default="/n/groups/marks/software/hhsuite/hhsuite-3.3.0", help="hhsuite root")),
    ("--scoring-window", dict(type=str, default="optimal", help="sequences longer than the context: optimal | overlapping")),
    ("--overwrite-prior-scores", dict(action="store_true", help="recompute columns already present in the output CSV")),
    ("--target_seq", dict(type=str, default=None, help="wild-type sequence (manual mode, no --dms_mapping)")),
    ("--weight_file_name", dict(type=str, default=None, help="sequence-weight file inside --msa-weights-folder (manual mode)")),
    ("--MSA_start", dict(type=int, default=None, help="first target position covered by the alignment, 1-indexed (manual mode)")),
    ("--MSA_end", dict(type=int, default=None, help="last target position covered by the alignment, 1-indexed (manual mode)")),
    ("--nogpu", dict(action="store_true", help="rejected: this scorer has no CPU path")),
    ("--device", dict(type=int, default=int(os.environ.get("LOCAL_RANK", "0")), help="[additive] GPU index (default LOCAL_RANK or 0)")),
    ("--precision", dict(type=str, default="f16x3", choices=sorted(pesm._lib.PRECISIONS),
                         help="[additive] f16x3 (default: split-fp16, 3 MFMAs per product, fp32-class accuracy, parity-gated), "
                              "fp32 (fp32 MFMA, parity-gated, slower), bf16 (fast, NOT parity-gated)")),
    ("--all-positions", dict(action="store_true", help="[additive] forward every token position as the reference does "
                                                       "(default: only positions some mutant reads; same outputs)")),
    ("--shard-positions", dict(action="store_true", help="[additive] MSA Transformer under torchrun: every rank forwards its share of "
                                                         "the masked positions of every seed, tables are all_gathered (see run_sharded)")),
]


def create_parser():
    parser = argparse.ArgumentParser(description="Zero-shot scoring of a deep mutational scan with ESM models on MI355X")
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    return parser


def label_rows(mutants, sequence, token_probs, offset_idx):
    """``label_row`` (compute_fitness.py:240-250) for a whole mutant column on a host table (token_probs [1, L+2, 33] or [L+2, 33]):
    the strings go through the library's parser (``pgmi_parse_mutants``: the reference's wild-type assertion, same text), the
    look-ups are ``esm.score_parsed`` -- an f32 difference per substitution added up in double in the order of the string, the
    arithmetic of the reference's ``.item()`` sum and of the device kernel.  Residue letters have the same indices in the ESM and
    the MSA Transformer alphabets (esm/data.py:142-174), so one parser serves both."""
    tp = token_probs[0] if token_probs.ndim == 3 else token_probs
    return pesm.score_from_table(tp, [str(m) for m in mutants], sequence, offset_idx)


def label_row(row, sequence, token_probs, alphabet, offset_idx):
    """compute_fitness.py:240-250, the seam function by name: one row of ``label_rows``."""
    return float(label_rows([row], sequence, token_probs, offset_idx)[0])


def mutated_sequences(mutants, wt_sequence, offset_idx):
    """``get_mutated_sequence`` (compute_fitness.py:252-257) for a column of SINGLE substitutions ('A25G'): wild-type copies as one
    byte matrix, one assignment.  A string that is not letter-integer-letter fails in ``int`` as in the reference.  Rows whose index
    falls BEFORE the sequence (a wrong --offset-idx: idx < 0) go through the reference's own expression
    ``wt[:idx] + mt + wt[idx + 1:]`` one by one -- python's negative indexing makes that a string of a different length (2 L for
    idx = -1), which the vectorised form cannot hold -- so malformed input gives the reference's output, not a silent in-place edit.
    (Difference kept: the whole column is parsed and validated before anything is built, so with several bad rows the FIRST error
    raised may be another row's than in the reference's row-by-row loop.)"""
    mutants = [str(m) for m in mutants]
    wt = np.frombuffer(wt_sequence.encode("ascii"), dtype=np.uint8)
    idx = np.array([int(m[1:-1]) for m in mutants], dtype=np.int64) - int(offset_idx)
    if ((idx >= len(wt)) | (idx < -len(wt))).any():
        raise IndexError("string index out of range")
    listed = np.frombuffer("".join(m[0] for m in mutants).encode("ascii"), dtype=np.uint8)
    assert (wt[idx] == listed).all(), "The listed wildtype does not match the provided sequence"
    out = np.tile(wt, (len(mutants), 1))
    out[np.arange(len(mutants)), idx] = np.frombuffer("".join(m[-1] for m in mutants).encode("ascii"), dtype=np.uint8)
    flat = out.tobytes().decode("ascii")
    seqs = [flat[i * len(wt):(i + 1) * len(wt)] for i in range(len(mutants))]
    for i in np.flatnonzero(idx < 0):                                  # the reference's expression, verbatim semantics
        k = int(idx[i])
        seqs[i] = wt_sequence[:k] + mutants[i][-1] + wt_sequence[(k + 1):]
    return seqs


def get_mutated_sequence(row, wt_sequence, offset_idx):
    """compute_fitness.py:252-257 (one single substitution)."""
    return mutated_sequences([row], wt_sequence, offset_idx)[0]


def compute_pppl_batch(sequences, model, alphabet):
    """compute_fitness.py:258-279 for a whole ``mutated_sequence`` column: for i in range(1, len(seq)-1) mask *token* i
    and read log p(sequence[i]) there (the reference's off-by-one and never-scored end residues are reproduced; no
    windowing, so ESM-1b raises above 1024 tokens).  The sequences are uploaded once (one byte per token) and every
    (sequence, i) row is enumerated on the device; mixed lengths (indels) share batches.  Host memory: O(total residues)."""
    library = pesm.SequenceLibrary(model, [str(s) for s in sequences], alphabet)
    try:
        return library.score()
    finally:
        library.close()


def compute_pppl_msa(sequence, model, alphabet, msa_rows):
    """compute_fitness.py:258-279 with mode == "MSA_Transformer", AS THE REFERENCE BEHAVES: the mutated sequence is put in front
    of the sampled alignment ([1, R+1, L+1] tokens), and for i in range(1, len(sequence) - 1) the statement
    ``batch_tokens_masked[0, i] = mask`` of :272 indexes the ROW axis of that 3-D tensor -- it masks alignment row i entirely
    (not token i of the first row) -- before log p(sequence[i]) is read at row 0, column i (:278).  Alignments with fewer than
    len(sequence) - 1 rows end in the reference's IndexError.  Reproduced as is (no launcher uses this branch; one full
    alignment-wide forward per residue and mutant, as in the reference)."""
    tokens = alphabet.get_batch_converter()([[("protein1", sequence)] + list(msa_rows)])[2][0]        # [R+1, L+1]
    n_rows = tokens.shape[0]
    total = []
    for i in range(1, len(sequence) - 1):
        if i >= n_rows:
            raise IndexError(f"index {i} is out of bounds for dimension 1 with size {n_rows}")
        masked = tokens.copy()
        masked[i, :] = alphabet.mask_idx
        total.append(float(model.token_logprobs(masked)[0, i, alphabet.get_idx(sequence[i])]))
    return sum(total)


def window_blend_weights():
    """Weight of a token by its place in a 1 024-token window when overlapping windows are averaged (compute_fitness.py:439-443):
    1 in the middle, a logistic ramp centred 128 tokens inside either edge (scale 16) over tokens 1..256 and 766..1022; the
    first and the last token of a window keep weight 1.  Evaluated in double, stored in float32, as the reference does."""
    w = np.ones(1024, dtype=np.float32)
    w[1:257] = [1 / (1 + math.exp(-(i - 128) / 16)) for i in range(1, 257)]
    w[766:1023] = [1 / (1 + math.exp((i - 894) / 16)) for i in range(766, 1023)]
    return w


def overlapping_windows(n_tok: int):
    """[(first, last)] token spans of the 1 024-token windows that cover a protein of n_tok > 1 024 tokens, in the order the
    reference accumulates them (compute_fitness.py:444-470): a left window from token 0 and a right window ending at the last
    token, both stepped inwards by 511 until they overlap; one more window around the middle when that last overlap is
    narrower than 511 tokens.  ``run_benchmark`` prices an assay by the length of this list."""
    left, right = 0, n_tok - 1024
    spans = [(left, left + 1023), (right, right + 1023)]
    while left + 1023 <= right:
        left, right = left + 511, right - 511
        spans += [(left, left + 1023), (right, right + 1023)]
    if left + 1023 - right + 1 < 511:
        centre = int(n_tok / 2) - 512
        spans.append((centre, centre + 1023))
    return spans


def wt_marginals_table(model, alphabet, sequence, scoring_window):
    """compute_fitness.py:433-475: the log-prob table of the unmasked wild type -- one forward, or with ``overlapping`` above
    1 024 tokens the weighted mean of the windows' tables (all windows go through the model as one batch; the sums run in
    float32 in the reference's order, so the blend has the reference's bits given the same window tables)."""
    _, _, batch_tokens = alphabet.get_batch_converter()([("protein1", sequence)])
    n_tok = batch_tokens.shape[1]
    if n_tok <= 1024 or scoring_window != "overlapping":
        return model.token_logprobs(batch_tokens)
    spans = overlapping_windows(n_tok)
    weights = window_blend_weights()
    tables = model.token_logprobs(np.stack([batch_tokens[0, a:b + 1] for a, b in spans]))
    total = np.zeros((1, n_tok, len(alphabet)), dtype=np.float32)
    norm = np.zeros((1, n_tok), dtype=np.float32)
    for (a, b), t in zip(spans, tables):
        total[:, a:b + 1] += t * weights.reshape(-1, 1)
        norm[:, a:b + 1] += weights
    return total / norm.reshape(1, -1, 1)


def _cell(row, name, default):
    """Value of an optional reference-file column ("" / NaN / missing column -> default)."""
    if name not in row.index:
        return default
    v = row[name]
    return default if (isinstance(v, float) and np.isnan(v)) or v == "" else v


def resolve_assay(args):
    """Everything main() needs about the assay, from the reference-file row (--dms_index) or from the manual flags:
    the rules of compute_fitness.py:286-340.  With the MSA Transformer the target sequence is cropped to the span the
    alignment covers and mutation positions are read relative to MSA_start (:318-325, :361).  Side effects kept from
    the reference: args.sequence / args.dms_input / args.dms_output / args.msa_path are rewritten."""
    wants_msa = "MSA_transformer" in args.model_type
    info = dict(mutant_col=args.mutation_col, msa_start=1, weight_file=None)
    if args.dms_index is not None:
        table = pd.read_csv(args.dms_mapping)
        dms_id = table["DMS_id"][args.dms_index]
        print("Compute scores for DMS: " + str(dms_id))
        hits = table[table["DMS_id"] == dms_id]
        if len(hits) != 1:
            raise ValueError(("No mappings found for DMS: " if len(hits) == 0 else "Multiple mappings found for DMS: ") + str(dms_id))
        row = hits.iloc[0]
        sequence = str(row["target_seq"]).upper()
        args.dms_input = str(args.dms_input) + os.sep + row["DMS_filename"]
        info["mutant_col"] = _cell(row, "DMS_mutant_column", args.mutation_col)
        first = int(_cell(row, "start_idx", 1))
        if wants_msa:
            name = _cell(row, "MSA_filename", "")
            if name == "":
                raise ValueError("No MSA found for DMS: " + str(dms_id))
            args.msa_path = str(args.msa_path) + os.sep + name       # --msa-path is the folder of alignments here
            m0, m1 = int(_cell(row, "MSA_start", 1)), int(_cell(row, "MSA_end", len(sequence)))
            if args.msa_weights_folder is not None and "weight_file_name" in row.index:
                info["weight_file"] = args.msa_weights_folder + os.sep + row["weight_file_name"]
            if first != m0 or first + len(sequence) != m1:            # the reference's (end-exclusive vs inclusive) test
                sequence, first = sequence[m0 - 1:m1], m0
            info["msa_start"] = m0
    else:
        dms_id = str(args.dms_input).split(os.sep)[-1].split(".csv")[0]
        sequence = args.target_seq.upper()
        first = args.offset_idx
        if args.MSA_start is None or args.MSA_end is None:
            if args.msa_path:
                print("MSA start and end not provided -- Assuming the MSA is covering the full WT sequence")
            args.MSA_start, args.MSA_end = 1, len(args.target_seq)
        info["msa_start"] = args.MSA_start
        if args.msa_weights_folder is not None and args.weight_file_name is not None:
            info["weight_file"] = args.msa_weights_folder + os.sep + args.weight_file_name
    args.sequence = sequence
    args.dms_output = str(args.dms_output) + os.sep + str(dms_id) + ".csv"
    info.update(dms_id=dms_id, first_position=first, frame=pd.read_csv(args.dms_input))
    return info


def checkpoint_stem(path):
    return path.split("/")[-1].split(".")[0]


def write_atomically(df, path):
    tmp = str(path) + ".tmp"
    df.to_csv(tmp, index=False)
    os.replace(tmp, path)                      # a crashed shard never leaves a partial CSV


def main(args):
    os.makedirs(args.dms_output, exist_ok=True)
    print("Arguments:", args)
    info = resolve_assay(args)
    df, mutant_col = info["frame"], info["mutant_col"]
    if len(df) == 0:
        raise ValueError("No rows found in the dataframe")
    print(f"df shape: {df.shape}", flush=True)
    if args.nogpu:
        raise RuntimeError("--nogpu: this scorer is GPU-only (libpgmi has no CPU path); "
                           "use the reference compute_fitness.py for CPU runs")
    print("Starting model scoring")
    if "MSA_transformer" in args.model_type:
        return score_msa_transformer(args, df, mutant_col, info["msa_start"], info["weight_file"])
    args.offset_idx = int(info["first_position"])
    needs_mutants = not (args.scoring_strategy == "pseudo-ppl" and "mutated_sequence" in df)
    mutants = [str(m) for m in df[mutant_col]] if needs_mutants else None
    columns = []
    for location in args.model_location:
        model, alphabet = pesm.load_model_and_alphabet(location, device=args.device, precision=args.precision)
        column = checkpoint_stem(location)
        columns.append(column)
        print("Transferred model to GPU")
        if args.scoring_strategy == "wt-marginals":            # one forward (or blended windows), then table look-ups
            table = wt_marginals_table(model, alphabet, args.sequence, args.scoring_window)
            df[column] = label_rows(mutants, args.sequence, table, args.offset_idx)
        elif args.scoring_strategy == "masked-marginals":      # the hot path: one device-resident assay
            print("Scoring with masked-marginals and model {}".format(column))
            if len(args.sequence) + 2 > 1024 and args.scoring_window == "overlapping":
                print("Overlapping not yet implemented for masked-marginals")
                sys.exit(0)
            assay = pesm.Assay(model, args.sequence, mutants, offset_idx=args.offset_idx, alphabet=alphabet,
                               window=1024, all_positions=args.all_positions)
            df[column] = assay.run()
            assay.close()
        else:                                                  # pseudo-ppl
            if "mutated_sequence" not in df:
                df["mutated_sequence"] = mutated_sequences(mutants, args.sequence, args.offset_idx)
            df[column] = compute_pppl_batch(list(df["mutated_sequence"]), model, alphabet)
        model.close()
    if "ESM1v" in args.model_type:                             # plain mean of the checkpoint columns (:530-537)
        df["Ensemble_ESM1v"] = sum(df[c] for c in columns) / len(columns)
    write_atomically(df, args.dms_output)


def score_msa_transformer(args, df, mutant_col, msa_start_index, MSA_weight_file_name):
    """MSA Transformer branch (compute_fitness.py:360-424, 538-543).  For every seed: sample ``--msa-samples`` rows of
    the pre-processed alignment (the wild type always first), run masked-marginals over the first row, add the column
    ``<checkpoint>_seed<k>``; finally ``<checkpoint>_ensemble`` = mean over the seeds.  Like the reference, the CSV is
    rewritten after every seed and seeds whose column already exists are skipped unless --overwrite-prior-scores."""
    from . import msa_transformer as pmsa
    seeds = list(args.seeds) if isinstance(args.seeds, (list, tuple)) else [args.seeds]
    assert args.scoring_strategy in ["masked-marginals", "pseudo-ppl"], "Zero-shot scoring strategy not supported with MSA Transformer"
    pppl = args.scoring_strategy == "pseudo-ppl"
    if pppl and getattr(args, "shard_positions", False):
        raise NotImplementedError("--shard-positions cuts the masked-marginals table; pseudo-ppl with the MSA Transformer runs unsharded")
    out_csv = args.dms_output
    shard_rank, shard_world = 0, 1
    if getattr(args, "shard_positions", False):
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized():
            shard_rank, shard_world = tdist.get_rank(), tdist.get_world_size()
    args.offset_idx = msa_start_index
    pad32 = lambda n: (n + 31) // 32 * 32
    for location in args.model_location:
        max_rows = pad32(min(args.msa_samples + (1 if pppl else 0), 1024)) * pad32(min(len(args.sequence) + 1, 1024))
        model, alphabet = pmsa.load_model_and_alphabet(location, device=args.device, max_rows=max_rows)
        stem = checkpoint_stem(location)
        print("Transferred model to GPU")
        to_tokens = alphabet.get_batch_converter()
        alignment = pmsa.process_msa(filename=str(args.msa_path), weight_filename=MSA_weight_file_name, filter_msa=args.filter_msa,
                                     path_to_hhfilter=args.path_to_hhfilter, hhfilter_min_cov=args.hhfilter_min_cov,
                                     hhfilter_max_seq_id=args.hhfilter_max_seq_id, hhfilter_min_seq_id=args.hhfilter_min_seq_id,
                                     device=args.device)
        for seed in seeds:
            column = f"{stem}_seed{seed}"
            on_disk = pd.read_csv(out_csv) if os.path.exists(out_csv) else None
            if on_disk is not None and column in on_disk.columns and not args.overwrite_prior_scores:
                print(f"Skipping seed {seed} as it is already in the dataframe")
                df = on_disk
                continue
            rows = pmsa.sample_msa(filename=str(args.msa_path), nseq=args.msa_samples, sampling_strategy=args.msa_sampling_strategy,
                                   random_seed=seed, weight_filename=MSA_weight_file_name, processed_msa=alignment,
                                   device=args.device)
            tokens = to_tokens([rows])[2][0]                      # [R, L+1]
            print(f"Batch sizes: {(1,) + tokens.shape}")
            if pppl:                                              # compute_fitness.py:403-417
                # row-wise from the CURRENT frame (compute_fitness.py:405-409): after an earlier seed or run it is the on-disk frame,
                # whose rows need not be the input file's in number or order
                if "mutated_sequence" not in df:
                    df["mutated_sequence"] = mutated_sequences(df[mutant_col], args.sequence, args.offset_idx)
                df[column] = [compute_pppl_msa(sq, model, alphabet, rows) for sq in df["mutated_sequence"]]
                if on_disk is not None and not args.overwrite_prior_scores:
                    assert column not in on_disk.columns, f"Column {column} already exists in {out_csv}"
                    df = on_disk.merge(df[[column, "mutant"]], on="mutant")
                df.to_csv(out_csv, index=False)
                continue
            T = tokens.shape[1]
            # the reference forwards every column; only cells some mutant reads are needed (--all-positions restores it)
            cells = sorted({1 + int(one[1:-1]) - args.offset_idx for m in df[mutant_col] for one in str(m).split(":")})   # +1: <cls>
            positions = list(range(T)) if args.all_positions else cells
            table = np.full((T, 33), np.nan, dtype=np.float32)
            # (seed, position) is the unit of work SURVEY 8e names for this path: with --shard-positions under torchrun every
            # rank samples the SAME rows (python RNG seeded by `seed`), forwards positions[rank::world] -- one alignment-wide
            # forward per masked column, all of equal cost -- and one all_gather + NaN-merge completes the table on every rank
            mine = positions[shard_rank::shard_world]
            if mine:
                table[mine] = model.masked_logprobs(tokens, mine, seq_len=len(args.sequence))
            if shard_world > 1:
                from . import dist as pdist
                import torch.distributed as tdist
                table = pdist.gather_tables({0: table}, [T], device="cuda" if tdist.get_backend() == "nccl" else "cpu")[0]
            df[column] = label_rows(df[mutant_col], args.sequence, table, args.offset_idx)   # the current frame's rows
            if on_disk is not None and not args.overwrite_prior_scores:
                assert column not in on_disk.columns, f"Column {column} already exists in {out_csv}"
                df = on_disk.merge(df[[column, "mutant"]], on="mutant")
            if shard_rank == 0:
                df.to_csv(out_csv, index=False)
            if shard_world > 1:
                import torch.distributed as tdist
                tdist.barrier()                           # the next seed re-reads the CSV rank 0 just wrote
        model.close()
    df[f"{stem}_ensemble"] = sum(df[f"{stem}_seed{seed}"] for seed in seeds) / len(seeds)
    if shard_rank == 0:
        write_atomically(df, out_csv)


if __name__ == "__main__":
    main(create_parser().parse_args())
