#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_esm.py tests/test_gpu_pppl.py tests/test_gpu_ops.py -x -q -m gpu > $O/test.log 2>&1; tail -3 $O/test.log
PGMI_BENCH_LEGS=esm2_3b,pseudo_ppl,tranception timeout 900 python bench.py --cpu-seconds 2 --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"])
for k,v in d.get("secondary",{}).items(): print(k, json.dumps(v)[:400])
PY
