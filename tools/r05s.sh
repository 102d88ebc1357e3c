#!/bin/bash
# round 5: shared-weights form of the default conv kernel (two m-tiles of one n-tile per 512-thread workgroup), A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05s; mkdir -p $O
for v in 1 2; do
  TISE_CONV_SHRB=$v timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or trunk" > $O/pytest_conv_shrb$v.txt 2>&1; tail -3 $O/pytest_conv_shrb$v.txt
done
for v in 0 1 2; do
  TISE_CONV_SHRB=$v timeout 300 python tools/split_layer_probe.py 500 > $O/layers_shrb$v.txt 2>&1
done
BENCH="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-host-feed --no-kernel-probe --png-images 0"
for rep in 1 2; do for v in 0 1 2; do
  TISE_CONV_SHRB=$v timeout 600 $BENCH > $O/bench_s${v}_$rep.json 2> $O/bench_s${v}_$rep.err
  python - <<PY
import json; d=json.load(open("$O/bench_s${v}_$rep.json")); print("shrb=$v rep $rep", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("fid"))
PY
done; done
