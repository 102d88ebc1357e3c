#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats run into a small committed summary.

usage: summarize_prof.py <dir with *_kernel_stats.csv and *_kernel_trace.csv> <out.md> [K steps]
Two tables: (1) the library's hand-written kernels over the whole run (calls, total, avg, min, max);
(2) steady-state per-step breakdown of the LAST K steps of the timed loop (window from the K-th last
resize kernel to the first pchol_init after it), which excludes MIOpen's find-mode trial kernels."""
import collections
import csv
import glob
import os
import sys

MINE = ("syrk_f32", "colsum", "stats_finalize", "resize_bilinear", "is_row", "is_col", "is_final", "pchol", "sytrd",
        "bisect", "gershgorin", "gemm_f64", "symmetrize", "frechet_finish", "axpy", "zero_rows", "bias_relu",
        "avgpool", "maxpool", "tise_")


def main():
    d, out = sys.argv[1], sys.argv[2]
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    stats = glob.glob(os.path.join(d, "*kernel_stats.csv"))[0]
    trace = glob.glob(os.path.join(d, "*kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    lines = ["# rocprofv3 --kernel-trace --stats summary", "", f"source: `{os.path.basename(stats)}`", "",
             "## hand-written kernels (whole run)", "",
             "| kernel | calls | total ms | avg us | min us | max us |", "|---|---:|---:|---:|---:|---:|"]
    for r in rows:
        if any(k in r["Name"] for k in MINE):
            name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
            lines.append(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                         f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} |")
    tr = []
    for r in csv.DictReader(open(trace)):
        tr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    tr.sort()
    res = [i for i, r in enumerate(tr) if "resize_bilinear" in r[2]]
    if len(res) >= K:
        i0 = res[-K]
        ends = [i for i, r in enumerate(tr) if i > i0 and "pchol_init" in r[2]]
        i1 = ends[0] if ends else len(tr)
        win = tr[i0:i1]
        span = (win[-1][1] - win[0][0]) / 1e6
        agg, cnt = collections.Counter(), collections.Counter()
        for s, e, n in win:
            if n.startswith("_ZN2ck") or "ck::" in n:
                key = "MIOpen: composable-kernel conv (ck::...grouped_conv_fwd...)"
            elif n.startswith("igemm"):
                key = "MIOpen: igemm_fwd_gtcx35_nhwc_fp32 asm conv"
            else:
                key = n.replace("(anonymous namespace)::", "").split("(")[0][:80]
            agg[key] += (e - s) / 1e6
            cnt[key] += 1
        lines += ["", f"## steady state: last {K} steps of the timed loop", "",
                  f"window {span:.2f} ms = {span/K:.3f} ms/step; sum of kernel durations {sum(agg.values()):.2f} ms", "",
                  "| kernel | ms/step | launches/step |", "|---|---:|---:|"]
        for k, v in agg.most_common(30):
            lines.append(f"| {k} | {v/K:.3f} | {cnt[k]/K:.1f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
