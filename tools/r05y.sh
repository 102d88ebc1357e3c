#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05y; mkdir -p $O
BENCH="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-host-feed --no-kernel-probe --png-images 0"
for rep in 1 2 3; do for v in 0 0x8000; do
  TISE_CONV_NSEG_FLAGS=$v timeout 600 $BENCH > $O/bench_f${v}_$rep.json 2> $O/bench_f${v}_$rep.err
  python - <<PY
import json; d=json.load(open("$O/bench_f${v}_$rep.json")); print("flags=$v rep $rep", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done; done
for v in 0 0x8000; do TISE_CONV_NSEG_FLAGS=$v timeout 300 python tools/split_layer_probe.py 3000 > $O/layers3000_f$v.txt 2>&1; done
