#!/bin/bash
# Round 6, the evidence under profiles/r6 from ONE box: the full -m gpu suite, the default bench line (secondary legs, in-run PMC traffic, end-to-end
# fields), the driver's command line, rocprofv3 kernel stats of the same command, the PMC passes (summary + traffic JSON stamped with the build digest),
# Tranception / MSA Transformer benches with kernel stats, attention by shape.       bash scripts/gpu/r6_final.sh [quick]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_final; rm -rf $O; mkdir -p $O
export PGMI_GIT_HEAD=${PGMI_GIT_HEAD:-unknown}
(rocm-smi --showpower --showclocks; rocm-smi --showmaxpower) > $O/box.txt 2>&1
python -c "import torch; p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, 'CUs', round(p.total_memory/2**30), 'GiB')" >> $O/box.txt 2>&1
if [ "$1" != "quick" ]; then
  timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; echo "rc $?" >> $O/gpu_suite.log; tail -4 $O/gpu_suite.log
fi
timeout 1200 python bench.py > $O/bench_f16x3.json 2> $O/bench_f16x3.err; echo "bench rc $?"
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary > $O/bench_driver_command_headline.json 2> $O/bench_driver.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f16x3 -o p -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary --no-box-state --no-live-traffic > $O/prof_f16x3.log 2>&1
if [ "$1" != "quick" ]; then
  bash scripts/pmc_profile.sh r6 --steps 1 --warmup 0 --cpu-seconds 0 --layers 4 --no-box-state --no-live-traffic > $O/pmc.log 2>&1; cp gpurun_out/pmc_r6/summary.txt $O/pmc_summary.txt; cp gpurun_out/pmc_r6/pmc_traffic.json $O/pmc_traffic.json
fi
timeout 400 python scripts/att_bench.py --rounds 7 --shapes 286x286,90x1100,150x150,600x120,200x230,120x500,60x737 --ab att_v3=0,att_v3=-1 > $O/att_bench.log 2>&1
# attention under the counters: the round-3 kernel and the software-pipelined one in the same process (separate passes per counter group)
mkdir -p $O/pmc_att
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_att/sq -o p -- python scripts/att_bench.py --rounds 2 --layers 2 --shapes 286x286,90x1100 --ab att_v3=0,att_v3=1 > $O/pmc_att/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_att/fetch -o p -- python scripts/att_bench.py --rounds 2 --layers 2 --shapes 286x286,90x1100 --ab att_v3=0,att_v3=1 > $O/pmc_att/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_att/valu -o p -- python scripts/att_bench.py --rounds 2 --layers 2 --shapes 286x286,90x1100 --ab att_v3=0,att_v3=1 > $O/pmc_att/valu.log 2>&1
python scripts/pmc_summarize.py $O/pmc_att 2>&1 | grep -A16 "attention_f16x3" > $O/pmc_attention_summary.txt; head -60 $O/pmc_attention_summary.txt
find $O/pmc_att -name "*.csv" -delete; find $O/pmc_att -name "*.db" -delete
timeout 300 python scripts/bench_tranception.py > $O/bench_tranception.json 2> $O/bench_tranception.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tranception -o p -- python scripts/bench_tranception.py > $O/prof_tranception.log 2>&1
timeout 300 python scripts/bench_msa_transformer.py > $O/bench_msa_transformer.json 2> $O/bench_msa.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.db" -delete
du -sh $O; ls $O
python - <<PY
import json
for f in ("bench_f16x3", "bench_driver_command_headline"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"].get("traffic"), d.get("parity"), d.get("value_end_to_end"),
              d.get("one_gpu_same_workload_mutants_per_s"), d.get("benchmark_217_end_to_end_mutants_per_s"))
        sec = d.get("secondary", {})
        for k in ("tranception_l_one_batch", "tranception_l_whole_assay_with_retrieval", "esm2_3b_one_assay", "esm2_650m_pseudo_ppl_capsd_shaped", "esm1v_5_checkpoint_ensemble"):
            if k in sec: print("  ", k, {kk: vv for kk, vv in sec[k].items() if isinstance(vv, (int, float))})
    except Exception as e:
        print(f, "unreadable", e)
PY
