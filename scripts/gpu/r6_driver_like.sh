#!/bin/bash
# What the driver runs at round end, on the final head: the -m gpu suite, smoke(), the default bench line.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_driver_like; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
SECONDS=0
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? in $SECONDS s"
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["traffic"], d.get("value_end_to_end"), d.get("benchmark_217_end_to_end_mutants_per_s"))
print("stdout lines (must be 1):", len(open("$O/bench_default.json").read().strip().splitlines()))
for k in ("bf16_throughput_mode", "tranception_217_projection", "indels_projection"):
    v = d["secondary"].get(k, {})
    print(k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if isinstance(vv, (int, float))})
print(d["kernels"]["attention"], d.get("parity"))
print(d["secondary"]["leg_seconds"], {k: round(v["mutants_per_s"]) for k, v in d["secondary"].items() if isinstance(v, dict) and "mutants_per_s" in v})
PY
