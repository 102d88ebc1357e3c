#!/bin/bash
# The README recipe under rocprofv3 --memory-copy-trace, several times: per run the feed line and the host->device copies' own durations
# (is a slow run made of slow copies, or of copies that start late?).   tools/cli_copy_trace.sh RUNS
RUNS=${1:-6}
ROOT=$(pwd); D=$(mktemp -d /tmp/tise_ct_XXXX)
python3 - "$D" <<'PY'
import sys, os
d = sys.argv[1]; sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
n = 30000; dev = torch.device("cuda", 0); os.makedirs(os.path.join(d, "png"))
data = torch.cat([bench.synth_images_device(i, min(i + 1000, n), dev, seed=0) for i in range(0, n, 1000)])
np.save(os.path.join(d, "px.npy"), data.cpu().numpy())
PY
python3 - "$D" <<'PY'
import sys, os
d = sys.argv[1]; sys.path.insert(0, os.getcwd())
import bench
from concurrent.futures import ProcessPoolExecutor
n = 30000; step = -(-n // 64)
with ProcessPoolExecutor(16) as ex:
    list(ex.map(bench._write_pngs, [(os.path.join(d, "px.npy"), a, min(a + step, n), os.path.join(d, "png")) for a in range(0, n, step)]))
PY
rm $D/px.npy
python3 -m tise_toolbox_amd.fid_score --batch-size 50 --path2 $D/png --save-stats $D/ref.npz --synthetic-weights > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $RUNS); do
  rm -rf /tmp/ct_$i
  PYTHONPATH="$ROOT" rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d /tmp/ct_$i -o t -- python3 -m tise_toolbox_amd.fid_score --batch-size 50 --path1 $D/ref.npz --path2 $D/png --synthetic-weights 2> /tmp/ct_$i.err > /dev/null
  grep "png feed" /tmp/ct_$i.err | cut -c1-60,180-330
  python3 - $i <<'PY'
import csv, glob, sys, statistics
i = sys.argv[1]
f = glob.glob(f"/tmp/ct_{i}/**/*memory_copy_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
big = [r for r in rows if r.get("Direction", "").endswith("HOST_TO_DEVICE") or "HOST_TO_DEVICE" in r.get("Direction", "")]
big = [r for r in big if 1_000_000 < (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 0 + 2_000_000] or big
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in big)
if d:
    print(f"   run {i}: {len(d)} H2D copies: median {statistics.median(d):.0f} us, p90 {d[int(len(d) * 0.9)]:.0f} us, max {d[-1]:.0f} us, sum {sum(d) / 1e3:.0f} ms")
k = glob.glob(f"/tmp/ct_{i}/**/*kernel_trace.csv", recursive=True)
if k:
    allk = list(csv.DictReader(open(k[0])))
    kr = [r for r in allk if "png_unfilter" in r["Kernel_Name"]]
    du = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in kr)
    if du:
        print(f"   run {i}: {len(du)} unfilter launches: median {statistics.median(du):.0f} us, max {du[-1]:.0f} us, sum {sum(du) / 1e3:.1f} ms")
    # how long does an unfilter launch wait AFTER the last copy of its device batch has landed?  (stream order: it may start at once)
    cp = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in big)
    import bisect
    ends = [e for _, e in cp]; starts = [s for s, _ in cp]
    waits = []
    for r in sorted(kr, key=lambda r: int(r["Start_Timestamp"])):
        ks = int(r["Start_Timestamp"])
        j = bisect.bisect_right(ends, ks) - 1
        if j >= 0:
            waits.append((ks - ends[j]) / 1e3)
    if waits:
        print(f"   run {i}: unfilter start - end of the last copy before it: median {statistics.median(waits):.0f} us, max {max(waits):.0f} us, sum {sum(waits) / 1e3:.1f} ms")
    # idle gaps of the copy engine between consecutive copies (end -> next start), beyond 1 ms
    gaps = [(starts[j + 1] - ends[j]) / 1e3 for j in range(len(cp) - 1)]
    big_gaps = [g for g in gaps if g > 1000]
    print(f"   run {i}: gaps between consecutive copies > 1 ms: {len(big_gaps)}, sum {sum(big_gaps) / 1e3:.0f} ms; first copy -> last copy {(ends[-1] - starts[0]) / 1e6:.0f} ms")
    # which queue ids do the trunk's kernels, the unfilter kernel use?
    q = {}
    for r in allk:
        name = "unfilter" if "png_unfilter" in r["Kernel_Name"] else ("conv" if "conv_" in r["Kernel_Name"] else None)
        if name:
            q.setdefault(name, set()).add(r.get("Queue_Id"))
    print(f"   run {i}: queue ids: {q}")
PY
done
rm -rf $D
