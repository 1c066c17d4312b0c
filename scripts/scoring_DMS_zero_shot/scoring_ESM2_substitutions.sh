#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_DMS_zero_shot/scoring_ESM2_substitutions.sh (same zero_shot_config.sh, same variables, same CSVs).
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${model_checkpoint:=/path/to/esm2_t33_650M_UR50D.pt}" "${dms_output_folder:=${DMS_output_score_folder_subs}/ESM2/${esm2_size:-650M}}"
pgmi_esm ESM2
