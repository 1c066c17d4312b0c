#!/bin/bash
# Round 4, everything under profiles/r4 from ONE box: the full -m gpu suite, the driver's bench line, rocprofv3 kernel stats of the same
# command, the PMC passes bench.py reads its traffic from, interleaved GEMM A/B, clock / power legs, Tranception and MSA Transformer
# benches with their kernel stats, the two-ranks-on-one-GPU rehearsal of the N > 1 line.      bash scripts/gpu/r4_final.sh [quick]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r4_final; rm -rf $O; mkdir -p $O
export PGMI_GIT_HEAD=${PGMI_GIT_HEAD:-unknown}
(rocm-smi --showpower --showclocks; rocm-smi --showmaxpower) > $O/box.txt 2>&1
python -c "import torch; p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, 'CUs', round(p.total_memory/2**30), 'GiB')" >> $O/box.txt 2>&1
if [ "$1" != "quick" ]; then
  timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; echo "rc $?" >> $O/gpu_suite.log; tail -4 $O/gpu_suite.log
fi
timeout 900 python bench.py > $O/bench_f16x3.json 2> $O/bench_f16x3.err; echo "bench rc $?"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary > $O/bench_driver_command_headline.json 2> $O/bench_driver.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f16x3 -o p -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary --no-box-state > $O/prof_f16x3.log 2>&1
if [ "$1" != "quick" ]; then
  bash scripts/pmc_profile.sh r4 > $O/pmc.log 2>&1; cp gpurun_out/pmc_r4/summary.txt $O/pmc_summary.txt; cp gpurun_out/pmc_r4/pmc_traffic.json $O/pmc_traffic.json
fi
timeout 300 python scripts/gemm_ab.py 5 0 1002 1008 > $O/gemm_ab_group_m.log 2>&1
timeout 300 python scripts/gemm_power_legs.py fc2 0 0 > $O/power_fc2.log 2>&1
timeout 300 python scripts/gemm_power_legs.py fc1 0 > $O/power_fc1.log 2>&1
timeout 300 python scripts/gemm_epilogue_cost.py > $O/gemm_epilogue_cost.log 2>&1
timeout 300 python scripts/att_bench.py --rounds 5 > $O/att_bench.log 2>&1
timeout 300 python scripts/bench_tranception.py > $O/bench_tranception.json 2> $O/bench_tranception.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tranception -o p -- python scripts/bench_tranception.py > $O/prof_tranception.log 2>&1
timeout 300 python scripts/bench_msa_transformer.py > $O/bench_msa_transformer.json 2> $O/bench_msa.err
timeout 300 python scripts/bench_msa_transformer.py --cols 1024 --positions 3 > $O/bench_msa_transformer_1024.json 2>> $O/bench_msa.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msat -o p -- python scripts/bench_msa_transformer.py --positions 3 > $O/prof_msat.log 2>&1
timeout 300 python scripts/bench_msa_weights.py --n 100000 --l 400 > $O/bench_msa_weights.json 2> $O/bench_msa_weights.err
PGMI_BENCH_SHARE_GPU=1 PGMI_BENCH_217_ASSAYS=24 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 4 --warmup 1 > $O/rehearsal_bench_n2_shared_gpu.json 2> $O/rehearsal.err
tools/store_drain > $O/store_drain.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.db" -delete
du -sh $O; ls $O
python - <<PY
import json
for f in ("bench_f16x3", "bench_driver_command_headline"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"].get("traffic"), d.get("parity"), d.get("one_gpu_same_workload_mutants_per_s"))
    except Exception as e:
        print(f, "unreadable", e)
PY
