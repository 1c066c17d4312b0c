#!/bin/bash
# short-assay groups: bit-identity tests (toy + real width) and the A/B on the short end of the 217-assay table
set -u
O=gpurun_out/r3_short; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_parity_real_width.py -q -s -k "short_assay" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/rc.txt
tail -5 $O/tests.log
timeout 600 python scripts/short_assay_ab.py > $O/ab.json 2> $O/ab.err; echo "ab rc=$?" | tee -a $O/rc.txt
cat $O/ab.json; tail -3 $O/ab.err
