#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05af; mkdir -p $O
TISE_CONV_NW6=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or trunk" > $O/pytest_conv_nw6.txt 2>&1; tail -3 $O/pytest_conv_nw6.txt
TISE_CONV_NW6=1 TISE_CONV_PREFER_TN3=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "trunk" > $O/pytest_trunk_nw6_tn3.txt 2>&1; tail -3 $O/pytest_trunk_nw6_tn3.txt
for v in "0 0" "1 0" "1 1"; do set -- $v
  TISE_CONV_NW6=$1 TISE_CONV_PREFER_TN3=$2 timeout 300 python tools/split_layer_probe.py 3000 > $O/layers3000_nw$1_tn3$2.txt 2>&1
done
BENCH="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-host-feed --no-kernel-probe --png-images 0"
for rep in 1 2; do for v in "0 0" "1 0" "1 1"; do set -- $v
  TISE_CONV_NW6=$1 TISE_CONV_PREFER_TN3=$2 timeout 600 $BENCH > $O/bench_nw$1_tn3$2_$rep.json 2> $O/bench_nw$1_tn3$2_$rep.err
  python - <<PY
import json; d=json.load(open("$O/bench_nw$1_tn3$2_$rep.json")); print("nw6=$1 prefer_tn3=$2 rep $rep", round(d["value"]), d["ms_per_step"], d["roofline"]["frac"])
PY
done; done
