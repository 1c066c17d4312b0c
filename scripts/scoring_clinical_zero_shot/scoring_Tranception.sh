#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_clinical_zero_shot/scoring_Tranception.sh (same zero_shot_config.sh, same variables, same CSVs):
# one clinical gene (DMS_index) with inference-time retrieval.
source "$(dirname "${BASH_SOURCE[0]}")/../scoring_DMS_zero_shot/_pgmi_env.sh"
: "${output_scores_folder:=${clinical_output_score_folder_subs}/Tranception/Tranception_L}"
pgmi_tranception "${clinical_reference_file_path_subs}" "${clinical_data_folder_subs}" --inference_time_retrieval \
    --MSA_folder "${clinical_MSA_data_folder_subs}" --MSA_weights_folder "${clinical_MSA_weights_folder_subs}"
