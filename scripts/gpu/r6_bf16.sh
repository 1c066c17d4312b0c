#!/bin/bash
# Round 6: the bf16 throughput mode through the persistent kernel: op-level numerics, the model tests, then the bench leg.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_bf16; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_esm.py -q -x -k "bf16" > $O/bf16_tests.log 2>&1; echo "rc $?" >> $O/bf16_tests.log; tail -8 $O/bf16_tests.log
PGMI_BENCH_LEGS=bf16_throughput_mode,tranception_217_projection,indels_projection timeout 900 python bench.py --steps 3 --warmup 1 --cpu-seconds 1 --no-live-traffic --no-box-state > $O/bench_legs.json 2> $O/bench_legs.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$O/bench_legs.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["kernels"]["attention"])
s = d["secondary"]
print(json.dumps(s.get("bf16_throughput_mode"), indent=1)[:900])
for k in ("tranception_217_projection", "indels_projection"):
    v = s.get(k) or {}
    print(k, {kk: vv for kk, vv in v.items() if isinstance(vv, (int, float))})
    for e in v.get("sample", []): print("   ", e)
print({k: v for k, v in s.items() if k.endswith("_error")}, s.get("leg_seconds"))
PY
