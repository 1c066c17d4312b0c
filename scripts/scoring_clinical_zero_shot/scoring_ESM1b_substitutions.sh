#!/bin/bash
# MI355X drop-in for ProteinGym's scripts/scoring_clinical_zero_shot/scoring_ESM1b_substitutions.sh: the 2 525 clinical genes with ONE
# resident checkpoint per GPU (run_benchmark --scoring-strategy wt-marginals --scoring-window overlapping) instead of one process per gene.
source "$(dirname "${BASH_SOURCE[0]}")/../scoring_DMS_zero_shot/_pgmi_env.sh"
: "${model_checkpoint:=/path/to/esm1b_t33_650M_UR50S.pt}" "${dms_output_folder:=${clinical_output_score_folder_subs}/ESM1b/}" "${PGMI_GPUS:=1}"
PGMI_MULTI_MODULE=proteingym_amd.run_benchmark pgmi_run proteingym_amd.run_benchmark --model-location ${model_checkpoint} --model_type ESM1b \
    --dms_mapping "${clinical_reference_file_path_subs}" --dms-input "${clinical_data_folder_subs}" --dms-output "${dms_output_folder}" \
    --scoring-strategy wt-marginals --scoring-window overlapping
