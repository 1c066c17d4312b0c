#!/bin/bash
# PMC passes for the dominant kernels (run on the GPU box via gpurun).  Counters are collected in
# their own rocprofv3 runs (no --stats / trace domains mixed in), one counter group per pass.
#   scripts/pmc_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ARGS=${@:-"--steps 1 --warmup 0 --cpu-seconds 0 --layers 4 --no-box-state"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# every dispatch of a kernel at the full row count: the per-dispatch means below must not mix in the kept-rows launches of the last layer
export PGMI_KEEP_ROWS=0
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
pass() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python bench.py $ARGS > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
pass sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
pass tcc TCC_HIT_sum TCC_MISS_sum
python scripts/pmc_summarize.py $OUT $OUT/pmc_traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
