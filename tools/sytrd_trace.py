#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of tools/frechet_probe.py: duration and start-to-start period of the
tridiagonalisation launches by column index (last full-rank solve in the trace)."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    if "sytrd_fused" in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
# split into solves: a solve is a run of launches; the full-rank ones have 2047 launches
runs, cur = [], []
for a, b in rows:
    if cur and a - cur[-1][1] > 200000:
        runs.append(cur); cur = []
    cur.append((a, b))
runs.append(cur)
full = [r for r in runs if len(r) >= 2040]
run = full[-1]
print(f"{len(runs)} solves in the trace; last full one: {len(run)} launches, {(run[-1][1] - run[0][0]) / 1e6:.2f} ms")
print("column   duration us   period us")
for k in range(0, len(run) - 1, 128):
    seg = run[k:k + 128]
    dur = sum(b - a for a, b in seg) / len(seg) / 1e3
    per = (seg[-1][0] - seg[0][0]) / max(1, len(seg) - 1) / 1e3
    print(f"{k:5d}    {dur:8.2f}     {per:8.2f}")
