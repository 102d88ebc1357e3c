#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes over tools/conv_traffic_target.py (vector-memory path counters) into one table per kernel family:
sums over the conv launches of the LAST forward pass.  Usage: vmem_pmc_summary.py <dir with pass_*/..._counter_collection.csv> <out.md>"""
import csv, glob, os, re, sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
per = defaultdict(lambda: defaultdict(float))       # kernel family -> counter -> sum
counters = []
for f in sorted(glob.glob(os.path.join(root, "pass_*", "*counter_collection.csv"))):
    rows = list(csv.DictReader(open(f)))
    convs = [r for r in rows if "conv_" in r["Kernel_Name"] or "stem_mfma" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in convs})
    n = len(ids) // 3                                # three forward passes: keep the last
    keep = set(ids[-n:])
    for r in convs:
        if int(r["Dispatch_Id"]) not in keep:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        per["ALL"][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] not in counters:
            counters.append(r["Counter_Name"])
with open(out, "w") as fh:
    fh.write("| kernel | " + " | ".join(counters) + " |\n|---|" + "---:|" * len(counters) + "\n")
    for k in sorted(per, key=lambda k: -per[k].get(counters[0], 0)):
        fh.write(f"| `{k}` | " + " | ".join(f"{per[k].get(c, 0):.4g}" for c in counters) + " |\n")
print(open(out).read())
