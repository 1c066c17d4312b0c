#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_DMS_zero_shot/scoring_MSA_transformer_substitutions.sh (same zero_shot_config.sh, same variables, same CSVs).
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${model_checkpoint:=/path/to/esm_msa1b_t12_100M_UR50S.pt}" "${dms_output_folder:=${DMS_output_score_folder_subs}/MSA_Transformer/}"
: "${scoring_strategy:=masked-marginals}" "${scoring_window:=optimal}" "${random_seeds:=1 2 3 4 5}" "${DMS_index:=0}"
: "${DMS_MSA_weights_for_MSA_Transformer_folder:=${DMS_MSA_weights_folder}/DMS_msa_weights_for_MSA_Transformer}"
pgmi_run proteingym_amd.compute_fitness --model-location ${model_checkpoint} --model_type MSA_transformer --dms_index "${DMS_index}" \
    --dms_mapping "${DMS_reference_file_path_subs}" --dms-input "${DMS_data_folder_subs}" --dms-output "${dms_output_folder}" \
    --scoring-strategy "${scoring_strategy}" --scoring-window "${scoring_window}" --msa-path "${DMS_MSA_data_folder}" \
    --msa-weights-folder "${DMS_MSA_weights_for_MSA_Transformer_folder}" --seeds ${random_seeds}
