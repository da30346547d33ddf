#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/profile_pmc.sh) per kernel: mean counter value per launch.
usage: tools/summarize_pmc.py gpurun_out/pmc_<tag> [kernel-substring]"""
import csv, glob, os, sys, json
root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "ugs_"
agg = {}
for path in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if want not in k:
            continue
        short = k.split("::")[-1].split("(")[0]
        d = agg.setdefault(short, {}).setdefault(r["Counter_Name"], [])
        d.append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out = {}
for k, cs in agg.items():
    out[k] = {c: {"mean": sum(v for v, _ in vals) / len(vals), "launches": len(vals), "mean_ns": sum(t for _, t in vals) / len(vals)} for c, vals in cs.items()}
print(json.dumps(out, indent=1))
