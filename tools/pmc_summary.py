#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_summary.json.
usage: pmc_summary.py <dir> <out.json>.  Corrections per MI355X_MICROARCH.md section HBM: counters are in KiB;
on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced read stream -> doubled for the
kernels whose reads are of that kind; WRITE_SIZE is exact for 16 B/lane stores, uncalibrated otherwise (kept raw)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
# syrk: the S tiles are read 8 B/lane (17.3 MB of the fetch) and only X is read 16 B/lane -> mixed, left raw
WIDE_READ = {"resize_bilinear_u8_kernel": True, "syrk_f32_upper_bk64_kernel": False, "syrk_f32_upper_kernel": False,
             "colsum_f32_kernel": False, "stats_finalize_kernel": False}
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].replace("void ", "")
        vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in vals.items():
    if not any(s in k for s in ("resize", "syrk", "colsum", "stats_finalize")):
        continue
    # drop the first launches (cold caches / first touch)
    fetch = c.get("FETCH_SIZE", [])[2:] or c.get("FETCH_SIZE", [])
    write = c.get("WRITE_SIZE", [])[2:] or c.get("WRITE_SIZE", [])
    fk = sum(fetch) / len(fetch) if fetch else None
    wk = sum(write) / len(write) if write else None
    corr = 2.0 if WIDE_READ.get(k, False) else 1.0
    e = {"launches": max(len(c.get("FETCH_SIZE", [])), len(c.get("WRITE_SIZE", []))),
         "FETCH_SIZE_KiB_raw": fk, "WRITE_SIZE_KiB_raw": wk, "fetch_correction": corr}
    if fk is not None and wk is not None:
        e["hbm_bytes_per_launch"] = (fk * corr + wk) * 1024.0
    res[k] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
