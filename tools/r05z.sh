#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05z; mkdir -p $O
BENCH="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-host-feed --no-kernel-probe --png-images 0"
for rep in 1 2; do for db in 250 500 1000 3000; do for v in 0 0x8000; do
  TISE_CONV_NSEG_FLAGS=$v timeout 600 $BENCH --device-batch $db > $O/bench_db${db}_f${v}_$rep.json 2> $O/bench_db${db}_f${v}_$rep.err
  python - <<PY
import json; d=json.load(open("$O/bench_db${db}_f${v}_$rep.json")); print("device batch $db flags=$v rep $rep", round(d["value"]), d["ms_per_step"], d["roofline"]["frac"])
PY
done; done; done
