#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 counter CSVs under a directory (tools/pmc_k1.sh)."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in root.rglob("*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in root.rglob("*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"].split("(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
out = {}
for name, counters in acc.items():
    if "k1_" not in name and "--all" not in sys.argv:
        continue
    n_disp = None
    entry = {}
    for cname, vals in counters.items():
        entry[cname] = sum(vals) / len(vals)
    if dur.get(name):
        entry["kernel_ms"] = sum(dur[name]) / len(dur[name])
    if "GRBM_GUI_ACTIVE" in entry and "kernel_ms" in entry:
        entry["effective_clock_GHz"] = entry["GRBM_GUI_ACTIVE"] / 8 / (entry["kernel_ms"] * 1e6)
    if "SQ_WAVE_CYCLES" in entry:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS"):
            if k in entry:
                entry[k + "/WAVE_CYCLES"] = entry[k] / entry["SQ_WAVE_CYCLES"]
    out[name] = entry
print(json.dumps(out, indent=1))
