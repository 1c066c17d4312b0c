#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_DMS_zero_shot/scoring_MSA_transformer_substitutions.sh (same zero_shot_config.sh, same variables, same CSVs).
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${model_checkpoint:=/path/to/esm_msa1b_t12_100M_UR50S.pt}" "${dms_output_folder:=${DMS_output_score_folder_subs}/MSA_Transformer/}"
pgmi_esm MSA_transformer --scoring-window "${scoring_window:-optimal}" --seeds ${random_seeds:-1 2 3 4 5} --msa-path "${DMS_MSA_data_folder}" \
    --msa-weights-folder "${DMS_MSA_weights_for_MSA_Transformer_folder:-${DMS_MSA_weights_folder}/DMS_msa_weights_for_MSA_Transformer}"
