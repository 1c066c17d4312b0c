#!/bin/bash
# Round 5: the loop-structure probe under rocprofv3 --pmc (counters in their own run, kernel trace only): MfmaUtil and wait cycles of the
# 8-wave ping-pong loop against the two-workgroups-per-CU loop, with and without the FC1-like epilogue.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r5_probe_pmc; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq -o p -- tools/mfma_duo 2 > $O/run.log 2>&1; echo "rc $?"
python - <<'PY'
import csv, glob, os
from collections import defaultdict
root = "gpurun_out/r5_probe_pmc"
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for f in glob.glob(os.path.join(root, "sq", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[k][1] += 1
with open(os.path.join(root, "summary.txt"), "w") as out:
    for k in sorted(agg):
        c = {n: s / m for n, (s, m) in agg[k].items()}
        d = dur[k][0] / max(dur[k][1], 1)
        gui = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
        line = (f"{k:40s} avg {d / 1e6:7.3f} ms  MfmaUtil {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(gui * 1024, 1) * 100:5.1f} % of active cycles  "
                f"clock {gui / max(d, 1):.2f} GHz  wait_inst_any / wave_cycles {c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f}  "
                f"active_inst_any / wave_cycles {c.get('SQ_ACTIVE_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
        print(line); out.write(line + "\n")
PY
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
