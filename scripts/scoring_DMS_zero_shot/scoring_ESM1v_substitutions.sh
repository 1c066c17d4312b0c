#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_DMS_zero_shot/scoring_ESM1v_substitutions.sh (same zero_shot_config.sh, same variables, same CSVs).
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${model_checkpoint:=${model_checkpoint1:-/path/to/esm1v_t33_650M_UR90S_1.pt} ${model_checkpoint2:-/path/to/esm1v_t33_650M_UR90S_2.pt} ${model_checkpoint3:-/path/to/esm1v_t33_650M_UR90S_3.pt} ${model_checkpoint4:-/path/to/esm1v_t33_650M_UR90S_4.pt} ${model_checkpoint5:-/path/to/esm1v_t33_650M_UR90S_5.pt}}"
: "${dms_output_folder:=${DMS_output_score_folder_subs}/ESM1v/}" "${scoring_strategy:=masked-marginals}" "${scoring_window:=optimal}" "${DMS_index:=0}"
pgmi_run proteingym_amd.compute_fitness --model-location ${model_checkpoint} --model_type ESM1v --dms_index "${DMS_index}" \
    --dms_mapping "${DMS_reference_file_path_subs}" --dms-input "${DMS_data_folder_subs}" --dms-output "${dms_output_folder}" \
    --scoring-strategy "${scoring_strategy}" --scoring-window "${scoring_window}"
