#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats run into a small committed summary.

usage: summarize_prof.py <dir with *_kernel_stats.csv and *_kernel_trace.csv, or a rocpd *_results.db> <out.md> [K steps]
Two tables: (1) the library's hand-written kernels over the whole run (calls, total, avg, min, max);
(2) steady-state per-step breakdown of the LAST K steps of the timed loop (window from the K-th last
resize kernel to the first stats_finalize after the last one), which excludes warm-up and the once-per-job tail."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    """Kernel name without the anonymous-namespace prefix, arguments or Itanium mangling."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:
        k = int(m.group(1))
        return name[m.end():m.end() + k]
    return name.replace("(anonymous namespace)::", "").split("(")[0]

MINE = ("syrk_f32", "colsum", "stats_finalize", "resize_bilinear", "is_row", "is_col", "is_final", "pchol", "sytrd",
        "bisect", "gershgorin", "gemm_f64", "symmetrize", "frechet_finish", "axpy", "zero_rows", "bias_relu",
        "avgpool", "maxpool", "tise_", "conv_split", "conv_pipe", "conv_win32", "conv_regw32", "conv_poolin", "stem_conv", "split_mean")


def main():
    d, out = sys.argv[1], sys.argv[2]
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dbs = glob.glob(os.path.join(d, "*_results.db"))
    if dbs and not glob.glob(os.path.join(d, "*kernel_stats.csv")):
        # rocprofv3's default output (rocpd sqlite): rebuild the two CSV views from the `kernels` view
        import sqlite3
        con = sqlite3.connect(dbs[0])
        disp = con.execute("select name, start, end from kernels").fetchall()
        per = collections.defaultdict(list)
        for n, s0, e0 in disp:
            per[n].append(e0 - s0)
        rows = [{"Name": n, "Calls": len(v), "TotalDurationNs": sum(v), "AverageNs": sum(v) / len(v), "MinNs": min(v),
                 "MaxNs": max(v)} for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))]
        stats = dbs[0]
        trace_rows = [{"Start_Timestamp": s0, "End_Timestamp": e0, "Kernel_Name": n} for n, s0, e0 in disp]
    else:
        stats = glob.glob(os.path.join(d, "*kernel_stats.csv"))[0]
        trace = glob.glob(os.path.join(d, "*kernel_trace.csv"))[0]
        rows = list(csv.DictReader(open(stats)))
        trace_rows = list(csv.DictReader(open(trace)))
    lines = ["# rocprofv3 --kernel-trace --stats summary", "", f"source: `{os.path.basename(stats)}`", "",
             "## hand-written kernels (whole run)", "",
             "| kernel | calls | total ms | avg us | min us | max us |", "|---|---:|---:|---:|---:|---:|"]
    for r in rows:
        if any(k in r["Name"] for k in MINE):
            name = short(r["Name"])
            lines.append(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                         f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} |")
    tr = []
    for r in trace_rows:
        tr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    tr.sort()
    res = [i for i, r in enumerate(tr) if "resize_bilinear" in r[2]]
    if len(res) >= K:
        i0 = res[-K]
        # end of the loop: the (mu, sigma) finalisation after the LAST resize (round 2: the pivoted Cholesky of the
        # reference side runs on a side stream DURING the loop, so pchol_* no longer marks the end)
        ends = [i for i, r in enumerate(tr) if i > res[-1] and "stats_finalize" in r[2]]
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
                key = short(n)[:80]
            agg[key] += (e - s) / 1e6
            cnt[key] += 1
        lines += ["", f"## steady state: last {K} device batches of the timed loop (one resize launch = one device batch: bench.py "
                  "cuts a rank's images into equal device batches of at most engine.DEVICE_BATCH_DEFAULT -- 5000 since round 6, 3000 in rounds 4-5, 1000 before -- whatever --steps is)", "",
                  f"window {span:.2f} ms = {span/K:.3f} ms per device batch; sum of kernel durations {sum(agg.values()):.2f} ms", "",
                  "| kernel | ms per device batch | launches per device batch |", "|---|---:|---:|"]
        for k, v in agg.most_common(30):
            lines.append(f"| {k} | {v/K:.3f} | {cnt[k]/K:.1f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
