#!/bin/bash
# Round 6: the bf16 throughput mode (persistent one-plane GEMM, attention on the split-fp16 pipe, bf16-plane context): tests, the bench leg, the per-kernel split.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_bf16b; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_esm.py -q -x -k "bf16 or small_head or head_dim_128 or kept_rows" > $O/bf16_tests.log 2>&1; echo "rc $?" >> $O/bf16_tests.log; tail -5 $O/bf16_tests.log
PGMI_BENCH_LEGS=bf16 timeout 900 python bench.py --steps 3 --warmup 1 --cpu-seconds 1 --no-live-traffic --no-box-state > $O/bench_bf16_leg.json 2> $O/bench_bf16_leg.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$O/bench_bf16_leg.json").read().strip().splitlines()[-1])
print(json.dumps(d["secondary"].get("bf16_throughput_mode"), indent=1)[:700])
PY
timeout 300 python bench.py --precision bf16 --steps 5 --warmup 2 --cpu-seconds 0 --no-secondary --no-live-traffic --no-box-state > $O/bench_bf16_headline_form.json 2>$O/err2; python -c "
import json; d=json.loads(open('$O/bench_bf16_headline_form.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(json.dumps(d['kernels']))"
