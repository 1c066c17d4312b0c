#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_DMS_zero_shot/scoring_Tranception_indels.sh (same zero_shot_config.sh, same variables, same CSVs):
# indels WITH retrieval -- every scored sequence is re-aligned to the family alignment by the Clustal Omega executable you point at.
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${output_scores_folder:=${DMS_output_score_folder_indels}/Tranception/Tranception_L}" "${clustal_omega_location:=/path/to/clustalo}"
pgmi_tranception "${DMS_reference_file_path_indels}" "${DMS_data_folder_indels}" --indel_mode --clustal_omega_location "${clustal_omega_location}" \
    --inference_time_retrieval --MSA_folder "${DMS_MSA_data_folder}" --MSA_weights_folder "${DMS_MSA_weights_folder}" --scoring_window optimal
