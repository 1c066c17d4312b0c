#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a ONE-GPU box: two ranks share the GPU, collectives over gloo (host staging).
# Not a measurement -- it checks that the torchrun launch, the weak leg, the strong-scaling leg and the JSON line all work.
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r3_rehearsal; mkdir -p $O
PGMI_BENCH_SHARE_GPU=1 PGMI_BENCH_217_ASSAYS=${ASSAYS:-24} timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 1 > $O/bench_n2.log 2> $O/bench_n2.err
echo "rc=$?" | tee $O/rc.txt
grep '^{' $O/bench_n2.log > $O/bench_n2.json; head -c 1500 $O/bench_n2.json; echo; tail -5 $O/bench_n2.err
