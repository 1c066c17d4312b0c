"""Drop-in for ``proteingym/baselines/tranception/score_tranception_proteingym.py`` on MI355X.

Same flags, same input resolution (reference file row or manual fields), same output file
``<output_scores_folder>/<DMS_id>.csv`` with the columns the reference writes
(mutated_sequence, avg_score_L_to_R, avg_score_R_to_L, avg_score [+ mutant for indels]) so that
``proteingym/merge.py`` (key ``mutated_sequence``, config.json:43) consumes it unchanged.

Reference: /root/reference/proteingym/baselines/tranception/score_tranception_proteingym.py:14-124.
Additive flag: --device.  Not supported (raise): --model_framework JAX, indel scoring *with*
retrieval (needs Clustal Omega re-alignment, msa_utils.update_retrieved_MSA_log_prior_indel).
"""
from __future__ import annotations

import argparse
import os

import pandas as pd

from . import tranception as ptr


# (flag, argparse keywords) -- the flag names, types and defaults are the reference's CLI contract
# (score_tranception_proteingym.py:19-45); help texts are ours.
_FLAGS = [
    ("--checkpoint", dict(type=str, help="Tranception checkpoint directory (HuggingFace layout: config.json + weights)")),
    ("--model_framework", dict(type=str, default="pytorch", help="kept for compatibility; only 'pytorch' maps to the HIP backend")),
    ("--batch_size_inference", dict(type=int, default=20, help="sequences per device call")),
    ("--DMS_reference_file_path", dict(type=str, default=None, help="reference CSV listing the assays (DMS_id, target_seq, DMS_filename, MSA_*)")),
    ("--DMS_index", dict(type=int, default=0, help="row of the reference CSV to score")),
    ("--target_seq", dict(type=str, default=None, help="wild-type sequence (manual mode, no reference CSV)")),
    ("--DMS_file_name", dict(type=str, default=None, help="assay CSV inside --DMS_data_folder (manual mode)")),
    ("--MSA_filename", dict(type=str, default=None, help="alignment (a2m) built on the wild type, inside --MSA_folder (manual mode)")),
    ("--MSA_weight_file_name", dict(type=str, default=None, help="sequence-weight .npy inside --MSA_weights_folder (manual mode, optional)")),
    ("--MSA_start", dict(type=int, default=None, help="first target position covered by the alignment, 1-indexed (manual mode)")),
    ("--MSA_end", dict(type=int, default=None, help="last target position covered by the alignment, 1-indexed (manual mode)")),
    ("--DMS_data_folder", dict(type=str, help="folder holding the assay CSVs")),
    ("--output_scores_folder", dict(type=str, default="./", help="where <DMS_id>.csv is written")),
    ("--deactivate_scoring_mirror", dict(action="store_true", help="score left-to-right only (default: average with the reversed sequence)")),
    ("--indel_mode", dict(action="store_true", help="the assay holds insertions/deletions (mutated_sequence column) instead of substitutions")),
    ("--scoring_window", dict(type=str, default="optimal", help="how sequences longer than the context are cropped: optimal | sliding")),
    ("--num_workers", dict(type=int, default=10, help="accepted for compatibility (no data-loader workers here)")),
    ("--inference_time_retrieval", dict(action="store_true", help="fuse the autoregressive log-probabilities with the alignment prior")),
    ("--retrieval_inference_weight", dict(type=float, default=0.6, help="alpha: weight of the alignment prior in the fusion")),
    ("--MSA_folder", dict(type=str, default=".", help="folder holding the alignments")),
    ("--MSA_weights_folder", dict(type=str, default=None, help="folder holding the sequence-weight files")),
    ("--clustal_omega_location", dict(type=str, default=None, help="Clustal Omega executable: indel scoring with retrieval re-aligns every sequence with it")),
    ("--device", dict(type=int, default=int(os.environ.get("LOCAL_RANK", "0")), help="[additive] GPU index")),
]


def create_parser():
    parser = argparse.ArgumentParser(description="Tranception scoring on MI355X")
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    return parser


def _join(folder, name):
    return None if folder is None else folder + os.sep + name


def resolve_inputs(args):
    """Assay id, wild type, assay file and (with retrieval) the alignment inputs, from the reference CSV row or from
    the manual flags -- the resolution order of score_tranception_proteingym.py:49-85 (MSA_start becomes 0-indexed)."""
    msa = None
    if args.DMS_reference_file_path:
        table = pd.read_csv(args.DMS_reference_file_path)
        dms_id = table["DMS_id"][args.DMS_index]
        print("Compute scores for DMS: " + str(dms_id))
        row = table[table["DMS_id"] == dms_id]
        wild_type = row["target_seq"].values[0].upper()
        assay_file = row["DMS_filename"].values[0]
        if args.inference_time_retrieval:
            weights = _join(args.MSA_weights_folder, row["weight_file_name"].values[0]) if args.MSA_weights_folder else None
            msa = (_join(args.MSA_folder, table["MSA_filename"][args.DMS_index]), weights,
                   int(row["MSA_start"].values[0]) - 1, int(row["MSA_end"].values[0]))
    else:
        wild_type, assay_file = args.target_seq, args.DMS_file_name
        dms_id = assay_file.split(".")[0]
        if args.inference_time_retrieval:
            weights = _join(args.MSA_weights_folder, args.MSA_weight_file_name) if args.MSA_weights_folder is not None else None
            msa = (_join(args.MSA_folder, args.MSA_filename), weights, args.MSA_start - 1, args.MSA_end)
    return dms_id, wild_type, assay_file, msa


def retrieval_arguments(args, wild_type, msa):
    """The retrieval dict of ``tranception.from_pretrained`` / ``build_retrieval`` for one assay (None without
    --inference_time_retrieval)."""
    if msa is None:
        return None
    out = dict(MSA_filename=msa[0], MSA_weight_file_name=msa[1], MSA_start=msa[2], MSA_end=msa[3],
               full_protein_length=len(wild_type), retrieval_inference_weight=args.retrieval_inference_weight)
    if args.indel_mode:                                   # score_tranception_proteingym.py:87-99: every sequence is re-aligned with Clustal Omega
        if not args.clustal_omega_location:
            raise ValueError("--indel_mode with --inference_time_retrieval needs --clustal_omega_location <Clustal Omega executable>")
        out.update(retrieval_aggregation_mode="aggregate_indel", clustal_omega_location=args.clustal_omega_location)
    return out


def main(args=None):
    args = create_parser().parse_args() if args is None else args
    if args.model_framework != "pytorch":
        raise NotImplementedError("only --model_framework pytorch has an MI355X backend")
    dms_id, wild_type, assay_file, msa = resolve_inputs(args)
    retrieval = retrieval_arguments(args, wild_type, msa)
    print("Model leverages both autoregressive and retrieval inference" if retrieval else "Model only uses autoregressive inference")
    model = ptr.from_pretrained(args.checkpoint, device=args.device, scoring_window=args.scoring_window, retrieval=retrieval)
    os.makedirs(args.output_scores_folder, exist_ok=True)
    out_csv = args.output_scores_folder + os.sep + dms_id + ".csv"
    assay = pd.read_csv(args.DMS_data_folder + os.sep + assay_file, low_memory=False)
    scores = model.score_mutants(DMS_data=assay, target_seq=wild_type, scoring_mirror=not args.deactivate_scoring_mirror,
                                 batch_size_inference=args.batch_size_inference, num_workers=args.num_workers,
                                 indel_mode=args.indel_mode)
    scores.to_csv(out_csv + ".tmp", index=False)
    os.replace(out_csv + ".tmp", out_csv)                 # atomic: a crashed run never leaves a partial CSV
    model.close()


if __name__ == '__main__':
    main()
