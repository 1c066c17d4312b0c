#!/bin/bash
# round 4, call 1: GEMM launch parameters (XCD start stagger, XCD-major tail, LDS-transposed fp32 epilogue) -- op tests, interleaved A/B,
# clock / power legs, the headline bench with the candidates
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4a; mkdir -p $O
rocm-smi --showpower --showclocks > $O/smi_idle.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/test_ops.log 2>&1; echo "ops rc $?" >> $O/test_ops.log
timeout 600 python scripts/gemm_ab.py 5 1000 1512 1288 1320 1384 1448 1800 1832 1896 1960 > $O/gemm_ab_1.log 2>&1
timeout 300 python scripts/gemm_ab.py 5 1000 1064 1128 1576 1640 1832 > $O/gemm_ab_2.log 2>&1      # stagger without the XCD-major tail
timeout 300 python scripts/gemm_power_legs.py fc2 1000 1832 1000 1832 > $O/power_fc2.log 2>&1
timeout 300 python scripts/gemm_power_legs.py out 1000 1832 > $O/power_out.log 2>&1
for v in 1000 1832 1512 1896; do
  timeout 400 python bench.py --no-secondary --cpu-seconds 0 --variant $v > $O/bench_v$v.json 2> $O/bench_v$v.err
done
tail -3 $O/test_ops.log; cat $O/gemm_ab_1.log $O/gemm_ab_2.log $O/power_fc2.log $O/power_out.log
for v in 1000 1832 1512 1896; do python - <<PY
import json
try:
    d = json.loads(open("$O/bench_v$v.json").read().strip().splitlines()[-1])
    print($v, d["value"], d["ms_per_step"], d["roofline"]["achieved"], {k: (v.get("tflops"), v.get("ms_per_step")) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print($v, "unreadable", e)
PY
done
