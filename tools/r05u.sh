#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05u; mkdir -p $O
BENCH="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-host-feed --no-kernel-probe --png-images 0"
for rep in 1 2 3; do for v in 0 1 3; do
  TISE_CONV_SHRB=$v timeout 600 $BENCH > $O/bench_s${v}_$rep.json 2> $O/bench_s${v}_$rep.err
  python - <<PY
import json; d=json.load(open("$O/bench_s${v}_$rep.json")); print("shrb=$v rep $rep", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done; done
