#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05ag; mkdir -p $O
BENCH="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-host-feed --no-kernel-probe --png-images 0"
for rep in 1 2 3; do for v in 3 2; do
  TISE_TN_192=$v timeout 600 $BENCH > $O/bench_tn$v_$rep.json 2> $O/bench_tn$v_$rep.err
  python - <<PY
import json; d=json.load(open("$O/bench_tn$v_$rep.json")); print("tn192=$v rep $rep", round(d["value"]), d["ms_per_step"], d["roofline"]["frac"])
PY
done; done
TISE_TN_192=2 timeout 300 python tools/split_layer_probe.py 3000 2>&1 | grep "192 k" | cut -c1-90
