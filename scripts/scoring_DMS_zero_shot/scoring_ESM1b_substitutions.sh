#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_DMS_zero_shot/scoring_ESM1b_substitutions.sh (same zero_shot_config.sh, same variables, same CSVs).
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${model_checkpoint:=/path/to/esm1b_t33_650M_UR50S.pt}" "${dms_output_folder:=${DMS_output_score_folder_subs}/ESM1b/}" "${scoring_strategy:=wt-marginals}"
pgmi_esm ESM1b --scoring-window "${scoring_window:-overlapping}"
