"""Summarise rocprofv3 --pmc CSV output per kernel name: mean counter value per dispatch.
    python scripts/pmc_summarize.py <dir> [traffic.json]
With a second argument the per-kernel fabric traffic (FETCH_SIZE x 2 per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE,
both in bytes per dispatch) is also written as JSON: bench.py reads it to fill `roofline.traffic`."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
traffic = {}
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for f in glob.glob(os.path.join(root, "sq", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[k][1] += 1
for k in sorted(agg, key=lambda k: -dur[k][0]):
    d = dur[k]
    print(f"== {k}  dispatches={d[1]}  avg_ns(under pmc)={d[0] / max(d[1], 1):.0f}")
    for c, (s, n) in sorted(agg[k].items()):
        print(f"   {c:34s} mean/dispatch = {s / n:.6g}")
    c = {c: s / n for c, (s, n) in agg[k].items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs; MFMA busy cycles summed over 256 CUs x 4 SIMDs
        gui = c['GRBM_GUI_ACTIVE'] / 8.0
        print(f"   -> MfmaUtil = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024) * 100:.1f}% of active cycles  "
              f"clock ~ {gui / max(d[0] / max(d[1], 1), 1):.2f} GHz")
    if "FETCH_SIZE" in c:
        print(f"   -> HBM read  ~ {c['FETCH_SIZE'] * 1024 * 2 / 1e6:.1f} MB/dispatch (FETCH_SIZE KB x2 gfx950 correction)")
    if "WRITE_SIZE" in c:
        print(f"   -> HBM write ~ {c['WRITE_SIZE'] * 1024 / 1e6:.1f} MB/dispatch (uncalibrated)")
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        traffic[k] = {"dispatches": d[1], "avg_ns_under_pmc": d[0] / max(d[1], 1),
                      "fetch_bytes": c.get("FETCH_SIZE", 0.0) * 1024 * 2, "write_bytes": c.get("WRITE_SIZE", 0.0) * 1024,
                      "l2_hit_rate": (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])) if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c else None}
if len(sys.argv) > 2:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from proteingym_amd import build_native
    head = os.environ.get("PGMI_GIT_HEAD")
    if not head:
        try:
            import subprocess
            head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
        except Exception:
            head = None
    json.dump({"lib_digest": build_native._digest(), "git_head": head,
               "rows_per_launch": int(os.environ.get("PGMI_PMC_ROWS", "82368")),
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts "
                         "128-byte requests as 64 bytes); bytes at the L2<->fabric boundary (Infinity Cache hits included)",
               "kernels": traffic}, open(sys.argv[2], "w"), indent=1)
