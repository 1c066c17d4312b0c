#!/bin/bash
# Round evidence on the GPU box: bench JSON lines + rocprofv3 --kernel-trace --stats summaries of the same
# commands.  Outputs under gpurun_out/evidence/ (copy what is to be judged into profiles/<round>/).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/evidence
rm -rf $OUT; mkdir -p $OUT
timeout 300 python bench.py > $OUT/bench_f16x3.log 2>&1; tail -1 $OUT/bench_f16x3.log > $OUT/bench_f16x3.json
timeout 300 python bench.py --precision fp32 --cpu-seconds 0 > $OUT/bench_fp32.log 2>&1; tail -1 $OUT/bench_fp32.log > $OUT/bench_fp32.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f16x3 -o p -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/prof_f16x3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fp32 -o p -- python bench.py --precision fp32 --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/prof_fp32.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_msat -o p -- python scripts/bench_msa_transformer.py --positions 3 > $OUT/prof_msat.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_msaw -o p -- python scripts/bench_msa_weights.py --n 100000 --l 400 --cpu-rows 2000 > $OUT/prof_msaw.log 2>&1
timeout 200 python scripts/bench_msa_weights.py --n 100000 --l 400 > $OUT/bench_msa_weights.log 2>&1
timeout 200 python scripts/bench_tranception.py > $OUT/bench_tranception.log 2>&1
find $OUT -name "*kernel_stats.csv" | head
# keep only the small summaries
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
