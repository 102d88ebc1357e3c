#!/bin/bash
# round 5: persistent-tile form of the default conv kernel, A/B against one workgroup per tile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or trunk" > $O/pytest_conv.txt 2>&1; tail -3 $O/pytest_conv.txt
for v in 0 1; do
  TISE_CONV_PERSIST=$v timeout 300 python tools/split_layer_probe.py 500 > $O/layers_persist$v.txt 2>&1
done
BENCH="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-host-feed --no-kernel-probe --png-images 0"
for rep in 1 2 3; do for v in 0 1; do
  TISE_CONV_PERSIST=$v timeout 600 $BENCH > $O/bench_p${v}_$rep.json 2> $O/bench_p${v}_$rep.err
  python - <<PY
import json; d=json.load(open("$O/bench_p${v}_$rep.json")); print("persist=$v rep $rep", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done; done
