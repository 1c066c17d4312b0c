#!/bin/bash
# Round 4, after a kernel change late in the round: the tests closest to the change, the PMC passes (bench.py takes roofline.traffic only
# from counters collected on the SAME build), then the driver's bench line with that traffic, its kernel stats and the default line.
#     bash scripts/gpu/r4_refresh.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r4_refresh; rm -rf $O; mkdir -p $O
export PGMI_GIT_HEAD=${PGMI_GIT_HEAD:-unknown}
timeout 120 python -m pytest tests/test_gpu_outliers.py tests/test_gpu_ops.py -q -m gpu -k "outlier or range_guard or non_finite or split_plane or half_tail" > $O/tests.log 2>&1; tail -2 $O/tests.log
bash scripts/pmc_profile.sh r4 > $O/pmc.log 2>&1; cp gpurun_out/pmc_r4/summary.txt $O/pmc_summary.txt; cp gpurun_out/pmc_r4/pmc_traffic.json $O/pmc_traffic.json
cp gpurun_out/pmc_r4/pmc_traffic.json profiles/r4/pmc_traffic.json
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary > $O/bench_driver_command_headline.json 2> $O/bench_driver.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f16x3 -o p -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary --no-box-state > $O/prof_f16x3.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.db" -delete
timeout 100 python scripts/bench_msa_transformer.py > $O/bench_msa_transformer.json 2> $O/bench_msa.err
timeout 400 python bench.py > $O/bench_f16x3.json 2> $O/bench_f16x3.err
python - <<PY
import json
for f in ("bench_driver_command_headline", "bench_f16x3"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"].get("traffic"), d.get("parity"), d.get("one_gpu_same_workload_mutants_per_s"))
    except Exception as e:
        print(f, "unreadable", e)
PY
