"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (tools/pmc_mfma.sh).

Normalisation (checked against the algorithmic MFMA count of enc4.3: 128 frames x 294,912 v_mfma_f32_32x32x2_f32 x 64
cycles = the counter to the digit):
  SQ_VALU_MFMA_BUSY_CYCLES = sum over the chip's SIMDs of the cycles its matrix pipe was busy (64 per 32x32x2 f32 MFMA);
  GRBM_GUI_ACTIVE          = sum over the 8 XCDs of the cycles the kernel was resident
  => mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs x 4 SIMDs x GRBM_GUI_ACTIVE / 8)       (1.0 = every pipe busy every cycle)
  effective clock = GRBM_GUI_ACTIVE / 8 / kernel duration  (MI355X_MICROARCH.md, DVFS; reads high on dispatches < 0.3 ms)
  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY are quad-cycles summed over waves: reported as fractions of WAVE_CYCLES.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(s in k for s in ("conv", "score", "wgrad")):
            continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[k][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3   # us
out = {}
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    us = sum(dur[k].values()) / len(dur[k])
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    e = {"launches": len(dur[k]), "avg_us_profiled": round(us, 1)}
    if gui:
        e["mfma_util"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (128.0 * gui), 4)
        e["effective_clock_GHz"] = round(gui / 8.0 / (us * 1e3), 3)
    if wc:
        e["wave_time_fractions"] = {"issuing": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                                    "issue_stall(mfma RAW/pipe)": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                                    "parked(waitcnt/barrier)": round(m.get("SQ_WAIT_ANY", 0) / wc, 3)}
    e["counters"] = {c: round(v) for c, v in sorted(m.items())}
    out[k] = e
print(json.dumps({"normalisation": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); see tools/pmc_mfma_summary.py",
                  "kernels": out}, indent=1))
