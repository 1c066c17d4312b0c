#!/bin/bash
# Extra PMC passes (issue / stall side) for the dominant kernels; same rules as pmc_profile.sh (counters only, own runs).
#   scripts/pmc_stalls.sh <tag>
set -u
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PGMI_KEEP_ROWS=0
OUT=gpurun_out/pmc_stalls_$TAG
mkdir -p $OUT
pass() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --layers 4 --no-secondary > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
pass sq SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
pass fifo SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_WAVES
pass base SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
python scripts/pmc_summarize.py $OUT > $OUT/summary.txt 2>&1
head -120 $OUT/summary.txt
