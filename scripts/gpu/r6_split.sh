#!/bin/bash
# Round 6: the two-launch form of sequences with 4 f + 1 / 4 f + 2 query tiles: bit tests, interleaved A/B, the bench line.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_split; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_tranception.py tests/test_gpu_esm.py -m gpu -q -x -k "attention or split or launch_options or prefix or batch" > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
timeout 900 python scripts/att_bench.py --rounds 7 --shapes 286x286,150x150,600x120,90x1100,320x320,410x410 --ab att_split=0,att_split=1 > $O/att_ab_5_split_launch.log 2>&1; echo "att_bench rc $?"; cat $O/att_ab_5_split_launch.log | tail -16
timeout 600 python bench.py --steps 10 --warmup 3 --no-secondary > $O/bench_split.json 2> $O/bench.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$O/bench_split.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_end_to_end")}, d["roofline"]["achieved"])
print({k: v for k, v in d.get("kernels", {}).items() if k in ("attention", "layernorm")})
PY
