#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_DMS_zero_shot/scoring_ESM1b_substitutions.sh (same zero_shot_config.sh, same variables, same CSVs).
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${model_checkpoint:=/path/to/esm1b_t33_650M_UR50S.pt}" "${dms_output_folder:=${DMS_output_score_folder_subs}/ESM1b/}"
: "${scoring_strategy:=wt-marginals}" "${scoring_window:=overlapping}" "${DMS_index:=0}"
pgmi_run proteingym_amd.compute_fitness --model-location ${model_checkpoint} --model_type ESM1b --dms_index "${DMS_index}" \
    --dms_mapping "${DMS_reference_file_path_subs}" --dms-input "${DMS_data_folder_subs}" --dms-output "${dms_output_folder}" \
    --scoring-strategy "${scoring_strategy}" --scoring-window "${scoring_window}"
