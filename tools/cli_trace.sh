#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the README recipe (fid_score from PNG files), to see what the feed's copies wait for.
#   tools/cli_trace.sh N_IMAGES OUT_DIR
set -e
N=${1:-12000}; OUT=${2:-gpurun_out/r06_cli_trace}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /tmp/tise_trace_XXXX)
python3 - "$N" "$D" "$ROOT" <<'PY'
import sys, os
n, d, root = int(sys.argv[1]), sys.argv[2], sys.argv[3]
sys.path.insert(0, root)
import numpy as np, torch, bench
from concurrent.futures import ProcessPoolExecutor
dev = torch.device("cuda", 0)
data = torch.cat([bench.synth_images_device(i, min(i + 1000, n), dev, seed=0) for i in range(0, n, 1000)])
np.save(os.path.join(d, "px.npy"), data.cpu().numpy())
PY
python3 - "$N" "$D" "$ROOT" <<'PY'
import sys, os
n, d, root = int(sys.argv[1]), sys.argv[2], sys.argv[3]
sys.path.insert(0, root)
import bench
from concurrent.futures import ProcessPoolExecutor
os.makedirs(os.path.join(d, "png"))
step = -(-n // 64)
with ProcessPoolExecutor(16) as ex:
    list(ex.map(bench._write_pngs, [(os.path.join(d, "px.npy"), a, min(a + step, n), os.path.join(d, "png")) for a in range(0, n, step)]))
PY
cd "$ROOT"
python3 -m tise_toolbox_amd.fid_score --batch-size 50 --path2 "$D/png" --save-stats "$D/ref.npz" --synthetic-weights > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
TISE_TIMING=1 PYTHONPATH="$ROOT" rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$ROOT/$OUT" -o cli -- python3 -m tise_toolbox_amd.fid_score --batch-size 50 --path1 "$D/ref.npz" --path2 "$D/png" --synthetic-weights 2>&1 | grep "tise" | tail -8
rm -rf "$D"
