# Sourced by the launchers in this folder -- drop-in replacements for ProteinGym's scripts/scoring_DMS_zero_shot/*.sh of the
# same names (reference: scripts/scoring_DMS_zero_shot/scoring_ESM1v_substitutions.sh:1-31 and siblings).  They read the SAME
# zero_shot_config.sh and the same variables a ProteinGym user edits (model_checkpoint, DMS_index, ...; a value already in the
# environment wins over the placeholder), and call this repository's MI355X scorer instead of proteingym/baselines/*.
#   cp scripts/scoring_DMS_zero_shot/*.sh <ProteinGym>/scripts/scoring_DMS_zero_shot/      (or run them in place with
#   ZERO_SHOT_CONFIG=<ProteinGym>/scripts/zero_shot_config.sh); PGMI_REPO points at this repository when the copies live elsewhere.
#   PGMI_LAUNCH_ECHO=1 prints the command line instead of running it (tests/test_host_logic.py checks it against both argument parsers).
_pgmi_here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
export PGMI_REPO="${PGMI_REPO:-$(cd "${_pgmi_here}/../.." && pwd)}"
export PYTHONPATH="${PGMI_REPO}${PYTHONPATH:+:${PYTHONPATH}}"
_pgmi_cfg="${ZERO_SHOT_CONFIG:-../zero_shot_config.sh}"
if [ ! -f "${_pgmi_cfg}" ]; then
    echo "zero_shot_config.sh not found at ${_pgmi_cfg}: run from ProteinGym's scripts/scoring_DMS_zero_shot or set ZERO_SHOT_CONFIG" >&2
    exit 2
fi
_pgmi_cfg_dir="$(cd "$(dirname "${_pgmi_cfg}")" && pwd)"
pushd "${_pgmi_cfg_dir}/scoring_DMS_zero_shot" > /dev/null 2>&1 || pushd "${_pgmi_cfg_dir}" > /dev/null   # the config's paths are relative to a launcher folder
source "${_pgmi_cfg_dir}/$(basename "${_pgmi_cfg}")"
for _v in DMS_reference_file_path_subs DMS_reference_file_path_indels clinical_reference_file_path_subs clinical_reference_file_path_indels; do
    if [ -n "${!_v}" ] && [ "${!_v#/}" = "${!_v}" ]; then export "${_v}=$(pwd)/${!_v}"; fi
done
popd > /dev/null
pgmi_run() {                                   # pgmi_run <python module> <args...>
    if [ -n "${PGMI_LAUNCH_ECHO}" ]; then printf '%s\n' "$@"; return 0; fi
    local mod="$1"; shift
    if [ "${PGMI_GPUS:-1}" -gt 1 ] && [ -n "${PGMI_MULTI_MODULE}" ]; then   # whole benchmark over the GPUs of this node (one process per GPU, RCCL)
        python -m torch.distributed.run --nnodes=1 --nproc-per-node "${PGMI_GPUS}" --master-addr 127.0.0.1 -m "${PGMI_MULTI_MODULE}" "$@"
    else
        python -m "${mod}" "$@"
    fi
}
# pgmi_esm <model_type> [more flags]: assay ${DMS_index} of the substitution benchmark through compute_fitness with ${model_checkpoint}
pgmi_esm() {
    local kind="$1"; shift
    pgmi_run proteingym_amd.compute_fitness --model_type "${kind}" --model-location ${model_checkpoint} --dms-output "${dms_output_folder}" \
        --dms_index "${DMS_index:=0}" --dms-input "${DMS_data_folder_subs}" --dms_mapping "${DMS_reference_file_path_subs}" \
        --scoring-strategy "${scoring_strategy:=masked-marginals}" "$@"
}
# pgmi_tranception <mapping csv> <data folder> [more flags]: assay ${DMS_index} through score_tranception_proteingym with ${checkpoint}
pgmi_tranception() {
    local mapping="$1" folder="$2"; shift 2
    pgmi_run proteingym_amd.score_tranception_proteingym --checkpoint "${checkpoint:=/path/to/Tranception_Large}" --DMS_index "${DMS_index:=0}" \
        --DMS_data_folder "${folder}" --DMS_reference_file_path "${mapping}" --output_scores_folder "${output_scores_folder}" "$@"
}
ESM1V_FIVE="${model_checkpoint1:-/path/to/esm1v_t33_650M_UR90S_1.pt} ${model_checkpoint2:-/path/to/esm1v_t33_650M_UR90S_2.pt} ${model_checkpoint3:-/path/to/esm1v_t33_650M_UR90S_3.pt} ${model_checkpoint4:-/path/to/esm1v_t33_650M_UR90S_4.pt} ${model_checkpoint5:-/path/to/esm1v_t33_650M_UR90S_5.pt}"
