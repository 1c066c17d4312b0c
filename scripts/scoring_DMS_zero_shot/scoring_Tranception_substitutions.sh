#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_DMS_zero_shot/scoring_Tranception_substitutions.sh (same zero_shot_config.sh, same variables, same CSVs).
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${output_scores_folder:=${DMS_output_score_folder_subs}/Tranception/Tranception_L}"
pgmi_tranception "${DMS_reference_file_path_subs}" "${DMS_data_folder_subs}" --inference_time_retrieval --MSA_folder "${DMS_MSA_data_folder}" --MSA_weights_folder "${DMS_MSA_weights_folder}"
