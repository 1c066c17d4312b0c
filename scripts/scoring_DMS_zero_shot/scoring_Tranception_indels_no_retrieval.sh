#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_DMS_zero_shot/scoring_Tranception_indels_no_retrieval.sh (same zero_shot_config.sh, same variables, same CSVs).
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${output_scores_folder:=${DMS_output_score_folder_indels}/Tranception_no_retrieval/Tranception_L}"
pgmi_tranception "${DMS_reference_file_path_indels}" "${DMS_data_folder_indels}" --indel_mode
