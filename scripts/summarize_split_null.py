#!/usr/bin/env python3
"""per-dispatch rows of the mapping kernel from the rocprofv3 passes of scripts/profile_split_null.py -> JSON:
dispatch 0 = warm-up, 1 = the single launch, 2.. = the split launches"""
import csv, glob, json, os, sys
root = sys.argv[1]
out = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "map_kernel" in r["Kernel_Name"]]
    per = {}
    for r in rows:
        per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
        per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, d in per.items():
        ids = sorted(d)
        vals = [d[i] for i in ids]
        out[c] = dict(single_launch=vals[1], split_launches_sum=sum(vals[2:]), nsplit=len(vals) - 2)
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "map_kernel" in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    if len(d) > 2:
        out["kernel_ms"] = dict(single_launch=d[1], split_launches_sum=sum(d[2:]), nsplit=len(d) - 2)
print(json.dumps(out, indent=1))
