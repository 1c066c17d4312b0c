#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_clinical_zero_shot/scoring_Tranception_indels.sh (same zero_shot_config.sh, same variables, same
# CSVs): clinical indels with retrieval, every scored sequence re-aligned by the Clustal Omega executable you point at.
source "$(dirname "${BASH_SOURCE[0]}")/../scoring_DMS_zero_shot/_pgmi_env.sh"
: "${output_scores_folder:=${clinical_output_score_folder_indels}/Tranception/Tranception_L}" "${clustal_omega_location:=/path/to/clustalo}"
pgmi_tranception "${clinical_reference_file_path_indels}" "${clinical_data_folder_indels}" --indel_mode --clustal_omega_location "${clustal_omega_location}" \
    --inference_time_retrieval --MSA_folder "${clinical_MSA_data_folder_indels}" --MSA_weights_folder "${clinical_MSA_weights_folder_indels}"
