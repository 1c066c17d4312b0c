#!/bin/bash
# The whole substitution benchmark with the ESM-1v ensemble on every GPU of this node: one process per GPU, assays LPT-sharded by
# run_benchmark, one RCCL all_gather of the score vectors (BASELINE north star).  Same config file, folders and CSVs as
# scoring_ESM1v_substitutions.sh run 217 times.   PGMI_GPUS=8 bash scoring_ESM1v_substitutions_all_assays.sh
source "$(dirname "${BASH_SOURCE[0]}")/_pgmi_env.sh"
: "${model_checkpoint:=${ESM1V_FIVE}}"
: "${dms_output_folder:=${DMS_output_score_folder_subs}/ESM1v/}" "${PGMI_GPUS:=8}"
PGMI_MULTI_MODULE=proteingym_amd.run_benchmark pgmi_run proteingym_amd.run_benchmark --model-location ${model_checkpoint} --model_type ESM1v \
    --dms_mapping "${DMS_reference_file_path_subs}" --dms-input "${DMS_data_folder_subs}" --dms-output "${dms_output_folder}"
