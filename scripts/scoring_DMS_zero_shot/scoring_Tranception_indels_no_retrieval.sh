#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_DMS_zero_shot/scoring_Tranception_indels_no_retrieval.sh (same zero_shot_config.sh, same variables, same CSVs).
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${checkpoint:=/path/to/Tranception_Large}" "${output_scores_folder:=${DMS_output_score_folder_indels}/Tranception_no_retrieval/Tranception_L}" "${DMS_index:=0}"
pgmi_run proteingym_amd.score_tranception_proteingym --checkpoint "${checkpoint}" --DMS_reference_file_path "${DMS_reference_file_path_indels}" \
    --DMS_data_folder "${DMS_data_folder_indels}" --DMS_index "${DMS_index}" --output_scores_folder "${output_scores_folder}" --indel_mode
