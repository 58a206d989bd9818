#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of scripts/profile_pmc.sh into one JSON (per kernel: mean duration, counters)."""
import csv, glob, json, os, sys
root = sys.argv[1]
out = {"kernels": {}}
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = n.split("(")[0]
    return n.replace("void ", "").strip()[-70:]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if "cmx::" in k:
            out["kernels"].setdefault(k, {})["avg_ms"] = float(r["AverageNs"]) / 1e6
            out["kernels"][k]["calls"] = int(r["Calls"])
for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    agg = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "cmx::" not in k:
            continue
        agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        out["kernels"].setdefault(k, {})["vgpr"] = int(r["VGPR_Count"])
        out["kernels"][k]["scratch"] = int(r["Scratch_Size"])
    for (k, c), v in agg.items():
        out["kernels"].setdefault(k, {}).setdefault("pmc", {})[c] = sum(v) / len(v)
print(json.dumps(out, indent=1))
