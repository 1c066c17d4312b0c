#!/bin/bash
# Round 6: the driver's 8-GPU command line rehearsed on ONE GPU (PGMI_BENCH_SHARE_GPU=1: ranks share cuda:0, gloo for the exchange --
# RCCL refuses two ranks on one device; labelled REHEARSAL in the line, never a measurement), the N-rank runners at world 8
# (byte-identical files vs one process), then the N = 1 line with its new fields.       bash scripts/gpu/r6_world8.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_world8; rm -rf $O; mkdir -p $O
export PGMI_GIT_HEAD=${PGMI_GIT_HEAD:-unknown}
PGMI_BENCH_SHARE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 8 --steps 4 --warmup 1 > $O/rehearsal_bench_n8_shared_gpu.json 2> $O/rehearsal_bench_n8.err; echo "bench n8 rc $?"
echo "stdout lines of the 8-rank bench (must be 1): $(wc -l < $O/rehearsal_bench_n8_shared_gpu.json)"
tail -c 1200 $O/rehearsal_bench_n8_shared_gpu.json; echo
PGMI_TEST_WORLD=8 timeout 1200 python -m pytest tests/test_gpu_two_ranks.py -q -m gpu > $O/tests_world8.log 2>&1; echo "world-8 runners rc $?"; tail -4 $O/tests_world8.log
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-secondary > $O/bench_n1_no_secondary.json 2> $O/bench_n1.err; echo "bench n1 rc $?"
python - <<PY
import json
d = json.loads(open("$O/bench_n1_no_secondary.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_end_to_end")}, d["roofline"]["achieved"], d["roofline"]["traffic"], str(d["roofline"]["traffic_detail"])[:400])
print(d.get("value_end_to_end_detail"))
PY
