"""Where the device's time goes in the from-PNG-files job: condenses a `rocprofv3 --kernel-trace` of tools/png_feed_probe.py.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_feed -- python3 tools/png_feed_probe.py 30000 14
    python3 tools/feed_trace_gaps.py /tmp/prof_feed
For every from-files job in the trace (a run of launches that contains png_unfilter_kernel) and for the resident job before it:
span first launch -> last end, the union of all kernels' busy intervals, the idle remainder, the summed duration of the unfilter
launches, how much of that ran while a trunk kernel was running, and the trunk kernels' summed duration."""
import csv
import glob
import sys


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def overlap(a, b):
    """total time intervals of a spend inside the union of b (both lists of (s, e))"""
    b = sorted(b)
    merged = []
    for s, e in b:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    tot, j = 0, 0
    for s, e in sorted(a):
        while j < len(merged) and merged[j][1] < s:
            j += 1
        k = j
        while k < len(merged) and merged[k][0] < e:
            tot += max(0, min(e, merged[k][1]) - max(s, merged[k][0]))
            k += 1
    return tot


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f[0]))]
    rows.sort(key=lambda r: r[1])
    # a job = the launches between two resize-free gaps of more than 20 ms; keep jobs with >= 20 resize launches
    jobs, cur = [], []
    for r in rows:
        if cur and r[1] - max(x[2] for x in cur[-50:]) > 20e6:
            jobs.append(cur)
            cur = []
        cur.append(r)
    jobs.append(cur)
    for job in jobs:
        n_resize = sum(1 for r in job if "resize_bilinear" in r[0])
        if n_resize < 10:
            continue
        unf = [(s, e) for nm, s, e in job if "png_unfilter" in nm]
        oth = [(s, e) for nm, s, e in job if "png_unfilter" not in nm]
        span = max(e for _, _, e in job) - job[0][1]
        busy = union([(s, e) for _, s, e in job])
        line = (f"{'from files' if unf else 'resident  '}: {n_resize:3d} device batches, span {span / 1e6:8.1f} ms, some kernel running {busy / 1e6:8.1f} ms, "
                f"idle {(span - busy) / 1e6:6.1f} ms, trunk + statistics kernels (sum of durations) {sum(e - s for s, e in oth) / 1e6:8.1f} ms")
        if unf:
            line += (f", png_unfilter {len(unf)} launches {sum(e - s for s, e in unf) / 1e6:6.1f} ms of which {overlap(unf, oth) / 1e6:6.1f} ms beside another kernel")
        print(line)


if __name__ == "__main__":
    main()
