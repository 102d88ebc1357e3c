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
    kr = [r for r in csv.DictReader(open(k[0])) if "png_unfilter" in r["Kernel_Name"]]
    du = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in kr)
    if du:
        print(f"   run {i}: {len(du)} unfilter launches: median {statistics.median(du):.0f} us, max {du[-1]:.0f} us, sum {sum(du) / 1e3:.1f} ms")
PY
done
rm -rf $D
