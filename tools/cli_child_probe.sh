set -e
D=$(mktemp -d /tmp/tise_cp_XXXX)
python3 - "$D" <<'PY'
import sys, os
d = sys.argv[1]
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
n = 30000
dev = torch.device("cuda", 0)
os.makedirs(os.path.join(d, "png"))
data = torch.cat([bench.synth_images_device(i, min(i + 1000, n), dev, seed=0) for i in range(0, n, 1000)])
np.save(os.path.join(d, "px.npy"), data.cpu().numpy())
PY
python3 - "$D" <<'PY'
import sys, os
d = sys.argv[1]
sys.path.insert(0, os.getcwd())
import bench
from concurrent.futures import ProcessPoolExecutor
n = 30000; step = -(-n // 64)
with ProcessPoolExecutor(16) as ex:
    list(ex.map(bench._write_pngs, [(os.path.join(d, "px.npy"), a, min(a + step, n), os.path.join(d, "png")) for a in range(0, n, step)]))
PY
rm $D/px.npy
python3 -m tise_toolbox_amd.fid_score --batch-size 50 --path2 $D/png --save-stats $D/ref.npz --synthetic-weights > /dev/null 2>&1
python3 tools/cli_child_probe.py $D/png $D/ref.npz
rm -rf $D
