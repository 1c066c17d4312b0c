#!/bin/bash
# On-box A/B of two builds of libpgmi.so (box-to-box clock variance is larger than most kernel changes):
#   scripts/lib_ab.sh <other.so> [rounds]      alternates bench.py between the in-tree library and <other.so>
set -u
OTHER=$1; ROUNDS=${2:-2}
cp proteingym_amd/libpgmi.so /tmp/lib_new.so
for r in $(seq $ROUNDS); do
  for which in new other; do
    if [ $which = new ]; then cp /tmp/lib_new.so proteingym_amd/libpgmi.so; else cp $OTHER proteingym_amd/libpgmi.so; fi
    timeout 200 python bench.py --cpu-seconds 0 --steps 6 --warmup 2 --no-secondary 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); k = d['kernels']
print('$which', 'ms/step %.1f' % d['ms_per_step'], ' '.join('%s %.2f' % (n, k[n]['ms_per_step']) for n in ('gemm_qkv', 'attention', 'gemm_out', 'gemm_fc1', 'gemm_fc2', 'layernorm')))"
  done
done
cp /tmp/lib_new.so proteingym_amd/libpgmi.so
