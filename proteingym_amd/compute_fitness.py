"""Drop-in for ``proteingym/baselines/esm/compute_fitness.py`` (ESM-1v / ESM-1b / ESM2 branch)
running on MI355X through libpgmi.so.

Same flags (including the reference's mixed ``-``/``_`` spellings), same input resolution, same
output file name, same columns, same ensemble rule -- so
``scripts/scoring_DMS_zero_shot/scoring_ESM1v_substitutions.sh`` / ``scoring_ESM2_substitutions.sh``
/ ``scoring_ESM1b_substitutions.sh`` only need their ``python .../compute_fitness.py`` line pointed
here, and ``proteingym/merge.py`` + ``performance_DMS_benchmarks.py`` consume the CSV unchanged.

Reference: /root/reference/proteingym/baselines/esm/compute_fitness.py
  create_parser :100-238   main :282-543   label_row :240-250   compute_pppl :258-279
Additive flags (not in the reference): --device, --precision, --all-positions.
The MSA-Transformer branch (:358-425) is out of scope and raises.
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


def create_parser():
    parser = argparse.ArgumentParser(
        description="Label a deep mutational scan with predictions from an ensemble of ESM-1v models."  # noqa
    )
    parser.add_argument("--model_type", type=str, help="MSA_transformer Vs ESM1v Vs ESM1b",
                        default="MSA_transformer", nargs="+")
    parser.add_argument("--model-location", type=str, nargs="+",
                        help="PyTorch model file OR name of pretrained model to download (see README for models)")
    parser.add_argument("--sequence", type=str, help="Base sequence to which mutations were applied")
    parser.add_argument("--dms-input", type=pathlib.Path, help="CSV file containing the deep mutational scan")
    parser.add_argument("--dms_index", type=int, help="Index of DMS in mapping file")
    parser.add_argument("--dms_mapping", type=str, help="Location of DMS_mapping")
    parser.add_argument("--mutation-col", type=str, default="mutant",
                        help="column in the deep mutational scan labeling the mutation as 'AiB'")
    parser.add_argument("--dms-output", type=pathlib.Path,
                        help="Output file containing the deep mutational scan along with predictions")
    parser.add_argument("--offset-idx", type=int, default=1,
                        help="Offset of the mutation positions in `--mutation-col`")
    parser.add_argument("--scoring-strategy", type=str, default="wt-marginals",
                        choices=["wt-marginals", "pseudo-ppl", "masked-marginals"], help="")
    parser.add_argument("--msa-path", type=pathlib.Path, help="path to MSA (required for MSA Transformer)")
    parser.add_argument("--msa-sampling-strategy", type=str, default='sequence-reweighting',
                        help="Strategy to sample sequences from MSA [sequence-reweighting|random|first_x_rows]")
    parser.add_argument("--msa-samples", type=int, default=400,
                        help="number of sequences to randomly sample from the MSA")
    parser.add_argument("--msa-weights-folder", type=str, default=None,
                        help="Folder with weights to sample MSA sequences in 'sequence-reweighting' scheme")
    parser.add_argument('--seeds', type=int, default=1, help='Random seed used during training', nargs="+")
    parser.add_argument('--filter-msa', action='store_true',
                        help='Whether to use hhfilter to filter input MSA before sampling')
    parser.add_argument('--hhfilter-min-cov', type=int, default=75, help='minimum coverage with query (%%)')
    parser.add_argument('--hhfilter-max-seq-id', type=int, default=90, help='maximum pairwise identity (%%)')
    parser.add_argument('--hhfilter-min-seq-id', type=int, default=0,
                        help='minimum sequence identity with query (%%)')
    parser.add_argument('--path-to-hhfilter', type=str,
                        default='/n/groups/marks/software/hhsuite/hhsuite-3.3.0', help='Path to hhfilter binaries')
    parser.add_argument('--scoring-window', type=str, default='optimal',
                        help='Approach to handle long sequences [optimal|overlapping]')
    parser.add_argument('--overwrite-prior-scores', action='store_true',
                        help='Whether to overwrite prior scores in the dataframe')
    # No ref file provided
    parser.add_argument('--target_seq', default=None, type=str, help='WT sequence mutated in the assay')
    parser.add_argument('--weight_file_name', default=None, type=str,
                        help='Wild type sequence mutated in the assay (to be provided if not using a reference file)')
    parser.add_argument('--MSA_start', default=None, type=int,
                        help='Index of first AA covered by the MSA relative to target_seq coordinates (1-indexing)')
    parser.add_argument('--MSA_end', default=None, type=int,
                        help='Index of last AA covered by the MSA relative to target_seq coordinates (1-indexing)')
    parser.add_argument("--nogpu", action="store_true", help="Do not use GPU even if available")
    # additive
    parser.add_argument("--device", type=int, default=int(os.environ.get("LOCAL_RANK", "0")),
                        help="[pgmi] GPU index (default LOCAL_RANK or 0)")
    parser.add_argument("--precision", type=str, default="f16x3", choices=sorted(pesm._lib.PRECISIONS),
                        help="[pgmi] GEMM arithmetic: f16x3 (default; split-fp16 3-pass, fp32-class accuracy, parity-gated), "
                             "fp32 (fp32 MFMA, parity-gated, slower), bf16 (fast, NOT parity-gated)")
    parser.add_argument("--all-positions", action="store_true",
                        help="[pgmi] forward every token position like the reference does (default: only "
                             "positions some mutant reads; outputs are identical)")
    return parser


def label_row(row, sequence, token_probs, alphabet, offset_idx):
    """compute_fitness.py:240-250 on a host table (token_probs [1, L+2, 33] or [L+2, 33])."""
    tp = token_probs[0] if token_probs.ndim == 3 else token_probs
    score = 0
    for mutation in row.split(":"):
        wt, idx, mt = mutation[0], int(mutation[1:-1]) - offset_idx, mutation[-1]
        assert sequence[idx] == wt, "The listed wildtype does not match the provided sequence"
        wt_encoded, mt_encoded = alphabet.get_idx(wt), alphabet.get_idx(mt)
        score += float(tp[1 + idx, mt_encoded] - tp[1 + idx, wt_encoded])     # add 1 for BOS
    return score


def get_mutated_sequence(row, wt_sequence, offset_idx):
    """compute_fitness.py:252-257 (single substitution, as the reference)."""
    wt, idx, mt = row[0], int(row[1:-1]) - offset_idx, row[-1]
    assert wt_sequence[idx] == wt, "The listed wildtype does not match the provided sequence"
    return wt_sequence[:idx] + mt + wt_sequence[(idx + 1):]


def compute_pppl_batch(sequences, model, alphabet):
    """compute_fitness.py:258-279 for many sequences: for i in range(1, len(seq)-1) mask *token* i
    and read log p(sequence[i]) there (the reference's off-by-one and the two never-scored
    trailing residues are reproduced; no windowing, so ESM-1b raises above 1024 tokens).
    All (sequence, i) rows of equal length go through pgmi_masked_logprobs in large batches."""
    out = np.zeros(len(sequences), dtype=np.float64)
    by_len = {}
    for n, s in enumerate(sequences):
        by_len.setdefault(len(s), []).append(n)
    conv = alphabet.get_batch_converter()
    for L, idxs in by_len.items():
        if L < 3:
            continue
        pos = np.arange(1, L - 1)
        _, _, toks = conv([("protein1", sequences[n]) for n in idxs])
        rows = np.repeat(toks, len(pos), axis=0)
        mpos = np.tile(pos, len(idxs))
        lp = model.masked_logprobs(rows, mpos).reshape(len(idxs), len(pos), -1)
        for j, n in enumerate(idxs):
            tgt = np.array([alphabet.get_idx(sequences[n][i]) for i in pos])
            vals = lp[j, np.arange(len(pos)), tgt]
            out[n] = sum(float(v) for v in vals)          # python float sum, like sum(log_probs)
    return out


def wt_marginals_table(model, alphabet, sequence, scoring_window):
    """compute_fitness.py:433-475."""
    _, _, batch_tokens = alphabet.get_batch_converter()([("protein1", sequence)])
    seq_len = batch_tokens.shape[1]
    if seq_len > 1024 and scoring_window == "overlapping":
        token_probs = np.zeros((1, seq_len, len(alphabet)), dtype=np.float32)
        token_weights = np.zeros((1, seq_len), dtype=np.float32)
        weights = np.ones(1024, dtype=np.float32)          # 1 for 256<=i<1022-256
        for i in range(1, 257):
            weights[i] = 1 / (1 + math.exp(-(i - 128) / 16))
        for i in range(1022 - 256, 1023):
            weights[i] = 1 / (1 + math.exp((i - 1022 + 128) / 16))
        start_left_window, end_left_window = 0, 1023
        start_right_window = (seq_len - 1) - 1024 + 1
        end_right_window = seq_len - 1
        windows = []
        while True:
            windows.append((start_left_window, end_left_window))
            windows.append((start_right_window, end_right_window))
            if end_left_window > start_right_window:
                break
            start_left_window += 511; end_left_window += 511
            start_right_window -= 511; end_right_window -= 511
        final_overlap = end_left_window - start_right_window + 1
        if final_overlap < 511:
            start_central_window = int(seq_len / 2) - 512
            windows.append((start_central_window, start_central_window + 1023))
        lps = model.token_logprobs(np.stack([batch_tokens[0, s:e + 1] for s, e in windows]))
        for (s, e), lp in zip(windows, lps):               # same accumulation order as the reference
            token_probs[:, s:e + 1] += lp * weights.reshape(-1, 1)
            token_weights[:, s:e + 1] += weights
        token_probs = token_probs / token_weights.reshape(1, -1, 1)
    else:
        token_probs = model.token_logprobs(batch_tokens)
    return token_probs


def main(args):
    if not os.path.exists(args.dms_output):
        os.mkdir(args.dms_output)
    print("Arguments:", args)

    mutant_col = args.mutation_col
    if args.dms_index is not None:
        mapping_protein_seq_DMS = pd.read_csv(args.dms_mapping)
        DMS_id = mapping_protein_seq_DMS["DMS_id"][args.dms_index]
        print("Compute scores for DMS: " + str(DMS_id))
        row = mapping_protein_seq_DMS[mapping_protein_seq_DMS["DMS_id"] == DMS_id]
        if len(row) == 0:
            raise ValueError("No mappings found for DMS: " + str(DMS_id))
        elif len(row) > 1:
            raise ValueError("Multiple mappings found for DMS: " + str(DMS_id))
        row = row.iloc[0]
        row = row.replace(np.nan, "")
        args.sequence = row["target_seq"].upper()
        args.dms_input = str(args.dms_input) + os.sep + row["DMS_filename"]
        mutant_col = row["DMS_mutant_column"] if "DMS_mutant_column" in mapping_protein_seq_DMS.columns else mutant_col
        args.dms_output = str(args.dms_output) + os.sep + DMS_id + '.csv'
        target_seq_start_index = row["start_idx"] if "start_idx" in mapping_protein_seq_DMS.columns and row["start_idx"] != "" else 1
        target_seq_end_index = target_seq_start_index + len(args.sequence)
        if "MSA_transformer" in args.model_type:                       # compute_fitness.py:310-325
            msa_filename = row["MSA_filename"]
            if msa_filename == "":
                raise ValueError("No MSA found for DMS: " + str(DMS_id))
            args.msa_path = str(args.msa_path) + os.sep + msa_filename
            msa_start_index = int(row["MSA_start"]) if "MSA_start" in mapping_protein_seq_DMS.columns else 1
            msa_end_index = int(row["MSA_end"]) if "MSA_end" in mapping_protein_seq_DMS.columns else len(args.sequence)
            MSA_weight_file_name = args.msa_weights_folder + os.sep + row["weight_file_name"] \
                if ("weight_file_name" in mapping_protein_seq_DMS.columns and args.msa_weights_folder is not None) else None
            if (target_seq_start_index != msa_start_index) or (target_seq_end_index != msa_end_index):
                args.sequence = args.sequence[msa_start_index - 1:msa_end_index]
                target_seq_start_index = msa_start_index
                target_seq_end_index = msa_end_index
        df = pd.read_csv(args.dms_input)
    else:
        DMS_id = str(args.dms_input).split(os.sep)[-1].split('.csv')[0]
        args.dms_output = str(args.dms_output) + os.sep + DMS_id + '.csv'
        target_seq_start_index = args.offset_idx
        args.sequence = args.target_seq.upper()
        if (args.MSA_start is None) or (args.MSA_end is None):             # compute_fitness.py:333-339
            if args.msa_path:
                print("MSA start and end not provided -- Assuming the MSA is covering the full WT sequence")
            args.MSA_start = 1
            args.MSA_end = len(args.target_seq)
        msa_start_index = args.MSA_start
        msa_end_index = args.MSA_end
        MSA_weight_file_name = args.msa_weights_folder + os.sep + args.weight_file_name \
            if (args.msa_weights_folder is not None and args.weight_file_name is not None) else None
        df = pd.read_csv(args.dms_input)

    if len(df) == 0:
        raise ValueError("No rows found in the dataframe")
    print(f"df shape: {df.shape}", flush=True)
    if args.nogpu:
        raise RuntimeError("--nogpu: this scorer is GPU-only (libpgmi has no CPU path); "
                           "use the reference compute_fitness.py for CPU runs")

    print("Starting model scoring")
    if "MSA_transformer" in args.model_type:
        return score_msa_transformer(args, df, mutant_col, msa_start_index, MSA_weight_file_name)
    for model_location in args.model_location:
        model, alphabet = pesm.load_model_and_alphabet(model_location, device=args.device,
                                                       precision=args.precision)
        model_location = model_location.split("/")[-1].split(".")[0]
        print("Transferred model to GPU")
        args.offset_idx = target_seq_start_index
        mutants = [str(m) for m in df[mutant_col]]

        if args.scoring_strategy == "wt-marginals":
            token_probs = wt_marginals_table(model, alphabet, args.sequence, args.scoring_window)
            df[model_location] = [label_row(m, args.sequence, token_probs, alphabet, args.offset_idx) for m in mutants]
        elif args.scoring_strategy == "masked-marginals":
            print("Scoring with masked-marginals and model {}".format(model_location))
            if len(args.sequence) + 2 > 1024 and args.scoring_window == "overlapping":
                print("Overlapping not yet implemented for masked-marginals")
                sys.exit(0)
            assay = pesm.Assay(model, args.sequence, mutants, offset_idx=args.offset_idx, alphabet=alphabet,
                               window=1024, all_positions=args.all_positions)
            df[model_location] = assay.run()
            assay.close()
        elif args.scoring_strategy == "pseudo-ppl":
            if 'mutated_sequence' not in df:
                df['mutated_sequence'] = [get_mutated_sequence(m, args.sequence, args.offset_idx) for m in mutants]
            df[model_location] = compute_pppl_batch(list(df['mutated_sequence']), model, alphabet)
        model.close()

    # compute_fitness.py:530-537: plain mean of the checkpoint columns
    if "ESM1v" in args.model_type:
        df["Ensemble_ESM1v"] = 0.0
        for model_location in args.model_location:
            model_location = model_location.split("/")[-1].split(".")[0]
            df["Ensemble_ESM1v"] += df[model_location]
        df["Ensemble_ESM1v"] /= len(args.model_location)
    tmp = str(args.dms_output) + ".tmp"
    df.to_csv(tmp, index=False)
    os.replace(tmp, args.dms_output)            # atomic: a crashed shard never leaves a partial CSV


def score_msa_transformer(args, df, mutant_col, msa_start_index, MSA_weight_file_name):
    """compute_fitness.py:360-424 + 538-543: per seed sample the alignment, masked-marginals over the first
    row, one ``<checkpoint>_seed<k>`` column per seed, then their mean in ``<checkpoint>_ensemble``."""
    from . import msa_transformer as pmsa
    seeds = args.seeds if isinstance(args.seeds, (list, tuple)) else [args.seeds]
    assert args.scoring_strategy in ["masked-marginals", "pseudo-ppl"], "Zero-shot scoring strategy not supported with MSA Transformer"
    if args.scoring_strategy == "pseudo-ppl":
        raise NotImplementedError("pseudo-ppl with the MSA Transformer is not built (the reference launcher uses masked-marginals)")
    for model_location in args.model_location:
        max_rows = (min(args.msa_samples, 1024) + 31) // 32 * 32 * ((min(len(args.sequence) + 1, 1024) + 31) // 32 * 32)
        model, alphabet = pmsa.load_model_and_alphabet(model_location, device=args.device, max_rows=max_rows)
        model_location = model_location.split("/")[-1].split(".")[0]
        print("Transferred model to GPU")
        batch_converter = alphabet.get_batch_converter()
        args.offset_idx = msa_start_index
        processed_msa = pmsa.process_msa(filename=str(args.msa_path), weight_filename=MSA_weight_file_name,
                                         filter_msa=args.filter_msa, device=args.device)
        mutants = [str(m) for m in df[mutant_col]]
        for seed in seeds:
            if os.path.exists(args.dms_output):
                prior_score_df = pd.read_csv(args.dms_output)
                if f"{model_location}_seed{seed}" in prior_score_df.columns and not args.overwrite_prior_scores:
                    print(f"Skipping seed {seed} as it is already in the dataframe")
                    df = prior_score_df
                    continue
            data = [pmsa.sample_msa(sampling_strategy=args.msa_sampling_strategy, filename=str(args.msa_path), nseq=args.msa_samples,
                                    weight_filename=MSA_weight_file_name, processed_msa=processed_msa, random_seed=seed,
                                    device=args.device)]
            _, _, batch_tokens = batch_converter(data)
            print(f"Batch sizes: {batch_tokens.shape}")
            T = batch_tokens.shape[2]
            # the reference forwards every column; only the cells some mutant reads are needed (--all-positions restores it)
            if args.all_positions:
                positions = list(range(T))
            else:
                positions = sorted({1 + int(mu[1:-1]) - args.offset_idx for m in mutants for mu in m.split(":")})
            rows = model.masked_logprobs(batch_tokens[0], positions, seq_len=len(args.sequence))
            token_probs = np.full((T, 33), np.nan, dtype=np.float32)
            token_probs[positions] = rows
            df[f"{model_location}_seed{seed}"] = [label_row(m, args.sequence, token_probs, alphabet, args.offset_idx) for m in mutants]
            if os.path.exists(args.dms_output) and not args.overwrite_prior_scores:
                prior_score_df = pd.read_csv(args.dms_output)
                assert f"{model_location}_seed{seed}" not in prior_score_df.columns, \
                    f"Column {model_location}_seed{seed} already exists in {args.dms_output}"
                prior_score_df = prior_score_df.merge(df[[f"{model_location}_seed{seed}", "mutant"]], on="mutant")
                prior_score_df.to_csv(args.dms_output, index=False)
                df = prior_score_df
            else:
                df.to_csv(args.dms_output, index=False)
        model.close()
    df[f"{model_location}_ensemble"] = 0.0
    for seed in seeds:
        df[f"{model_location}_ensemble"] += df[f"{model_location}_seed{seed}"]
    df[f"{model_location}_ensemble"] /= len(seeds)
    tmp = str(args.dms_output) + ".tmp"
    df.to_csv(tmp, index=False)
    os.replace(tmp, args.dms_output)


if __name__ == "__main__":
    main(create_parser().parse_args())
