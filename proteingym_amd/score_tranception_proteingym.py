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


def create_parser():
    parser = argparse.ArgumentParser(description='Tranception scoring')
    parser.add_argument('--checkpoint', type=str, help='Path of Tranception model checkpoint')
    parser.add_argument('--model_framework', default='pytorch', type=str, help='Underlying framework [pytorch|JAX]')
    parser.add_argument('--batch_size_inference', default=20, type=int, help='Batch size for inference')
    parser.add_argument('--DMS_reference_file_path', default=None, type=str, help='Path to reference file with list of DMS to score')
    parser.add_argument('--DMS_index', default=0, type=int, help='Index of DMS assay in reference file')
    parser.add_argument('--target_seq', default=None, type=str, help='Full wild type sequence that is mutated in the DMS asssay')
    parser.add_argument('--DMS_file_name', default=None, type=str, help='Name of DMS assay file')
    parser.add_argument('--MSA_filename', default=None, type=str, help='Name of MSA (eg., a2m) file constructed on the wild type sequence')
    parser.add_argument('--MSA_weight_file_name', default=None, type=str, help='Weight of sequences in the MSA (optional)')
    parser.add_argument('--MSA_start', default=None, type=int, help='Sequence position that the MSA starts at (1-indexing)')
    parser.add_argument('--MSA_end', default=None, type=int, help='Sequence position that the MSA ends at (1-indexing)')
    parser.add_argument('--DMS_data_folder', type=str, help='Path to folder that contains all DMS assay datasets')
    parser.add_argument('--output_scores_folder', default='./', type=str, help='Name of folder to write model scores to')
    parser.add_argument('--deactivate_scoring_mirror', action='store_true', help='Whether to deactivate sequence scoring from both directions (Left->Right and Right->Left)')
    parser.add_argument('--indel_mode', action='store_true', help='Flag to be used when scoring insertions and deletions. Otherwise assumes substitutions')
    parser.add_argument('--scoring_window', default="optimal", type=str, help='Sequence window selection mode (when sequence length longer than model context size)')
    parser.add_argument('--num_workers', default=10, type=int, help='Number of workers for model scoring data loader')
    parser.add_argument('--inference_time_retrieval', action='store_true', help='Whether to perform inference-time retrieval')
    parser.add_argument('--retrieval_inference_weight', default=0.6, type=float, help='Coefficient (alpha) used when aggregating autoregressive transformer and retrieval')
    parser.add_argument('--MSA_folder', default='.', type=str, help='Path to MSA for neighborhood scoring')
    parser.add_argument('--MSA_weights_folder', default=None, type=str, help='Path to MSA weights for neighborhood scoring')
    parser.add_argument('--clustal_omega_location', default=None, type=str, help='Path to Clustal Omega (only needed with scoring indels with retrieval)')
    parser.add_argument('--device', type=int, default=int(os.environ.get("LOCAL_RANK", "0")), help='[pgmi] GPU index')
    return parser


def main(args=None):
    args = create_parser().parse_args() if args is None else args
    if args.model_framework != "pytorch":
        raise NotImplementedError("only --model_framework pytorch has an MI355X backend")
    if args.DMS_reference_file_path:
        mapping = pd.read_csv(args.DMS_reference_file_path)
        DMS_id = mapping["DMS_id"][args.DMS_index]
        print("Compute scores for DMS: " + str(DMS_id))
        row = mapping[mapping["DMS_id"] == DMS_id]
        target_seq = row["target_seq"].values[0].upper()
        DMS_file_name = row["DMS_filename"].values[0]
        if args.inference_time_retrieval:
            MSA_data_file = args.MSA_folder + os.sep + mapping["MSA_filename"][args.DMS_index] if args.MSA_folder is not None else None
            MSA_weight_file_name = args.MSA_weights_folder + os.sep + row["weight_file_name"].values[0] if args.MSA_weights_folder else None
            MSA_start = int(row["MSA_start"].values[0]) - 1
            MSA_end = int(row["MSA_end"].values[0])
    else:
        target_seq = args.target_seq
        DMS_file_name = args.DMS_file_name
        DMS_id = DMS_file_name.split(".")[0]
        if args.inference_time_retrieval:
            MSA_data_file = args.MSA_folder + os.sep + args.MSA_filename if args.MSA_folder is not None else None
            MSA_weight_file_name = args.MSA_weights_folder + os.sep + args.MSA_weight_file_name if args.MSA_weights_folder is not None else None
            MSA_start = args.MSA_start - 1
            MSA_end = args.MSA_end

    retrieval = None
    if args.inference_time_retrieval:
        if args.indel_mode:
            raise NotImplementedError("indel scoring with retrieval needs Clustal Omega re-alignment (not built)")
        retrieval = dict(MSA_filename=MSA_data_file, MSA_start=MSA_start, MSA_end=MSA_end, full_protein_length=len(target_seq),
                         retrieval_inference_weight=args.retrieval_inference_weight, MSA_weight_file_name=MSA_weight_file_name)
        print("Model leverages both autoregressive and retrieval inference")
    else:
        print("Model only uses autoregressive inference")
    model = ptr.from_pretrained(args.checkpoint, device=args.device, scoring_window=args.scoring_window, retrieval=retrieval)

    if not os.path.isdir(args.output_scores_folder):
        os.mkdir(args.output_scores_folder)
    scoring_filename = args.output_scores_folder + os.sep + DMS_id + ".csv"
    DMS_data = pd.read_csv(args.DMS_data_folder + os.sep + DMS_file_name, low_memory=False)
    all_scores = model.score_mutants(DMS_data=DMS_data, target_seq=target_seq,
                                     scoring_mirror=not args.deactivate_scoring_mirror,
                                     batch_size_inference=args.batch_size_inference, num_workers=args.num_workers,
                                     indel_mode=args.indel_mode)
    tmp = scoring_filename + ".tmp"
    all_scores.to_csv(tmp, index=False)
    os.replace(tmp, scoring_filename)
    model.close()


if __name__ == '__main__':
    main()
