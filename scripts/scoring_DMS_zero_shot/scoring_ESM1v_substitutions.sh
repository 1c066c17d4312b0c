#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_DMS_zero_shot/scoring_ESM1v_substitutions.sh (same zero_shot_config.sh, same variables, same CSVs).
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${model_checkpoint:=${ESM1V_FIVE}}" "${dms_output_folder:=${DMS_output_score_folder_subs}/ESM1v/}"
pgmi_esm ESM1v --scoring-window "${scoring_window:-optimal}"
