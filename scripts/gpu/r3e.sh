mkdir -p gpurun_out/r3e
for i in 1 2; do
  PGMI_ATT_TUNE=14 PGMI_ATT_DEFER=0 python bench.py --no-secondary --cpu-seconds 0 --steps 6 --warmup 2 > gpurun_out/r3e/bench_old_$i.json 2>/dev/null
  python bench.py --no-secondary --cpu-seconds 0 --steps 6 --warmup 2 > gpurun_out/r3e/bench_new_$i.json 2>/dev/null
done
python scripts/att_bench.py --rounds 5 --shapes 286x286,90x1100 > gpurun_out/r3e/att.log 2>&1
python -m pytest tests/test_gpu_msa_transformer.py tests/test_gpu_esm.py tests/test_gpu_ops.py -m gpu -q -x > gpurun_out/r3e/tests.log 2>&1
echo "rc_tests=$?" > gpurun_out/r3e/rc.txt
python - <<PY
import json
for n in ("old_1","new_1","old_2","new_2"):
    b=json.load(open("gpurun_out/r3e/bench_%s.json"%n)); k=b["kernels"]
    print(n, round(b["ms_per_step"],2), "att", k["attention"]["ms_per_step"], "fc2", k["gemm_fc2"]["ms_per_step"], "qkv", k["gemm_qkv"]["ms_per_step"])
PY
cat gpurun_out/r3e/rc.txt; grep -v "^{" gpurun_out/r3e/att.log | tail -8; tail -2 gpurun_out/r3e/tests.log
