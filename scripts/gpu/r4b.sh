#!/bin/bash
# round 4, call 2: fp32 epilogue with pipelined residual loads, first K tile of the next item in flight for every output kind
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/test_ops.log 2>&1; echo "ops rc $?" >> $O/test_ops.log
timeout 600 python scripts/gemm_ab.py 5 1000 1512 > $O/gemm_ab_1.log 2>&1
for v in 1000 0; do
  timeout 400 python bench.py --no-secondary --cpu-seconds 0 --variant $v > $O/bench_v$v.json 2> $O/bench_v$v.err
done
tail -3 $O/test_ops.log; cat $O/gemm_ab_1.log
for v in 1000 0; do python - <<PY
import json
try:
    d = json.loads(open("$O/bench_v$v.json").read().strip().splitlines()[-1])
    print($v, d["value"], d["ms_per_step"], d["roofline"]["achieved"], {k: (v.get("tflops"), v.get("ms_per_step")) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print($v, "unreadable", e)
PY
done
