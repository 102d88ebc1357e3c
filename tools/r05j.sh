#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or trunk" > $O/pytest_conv.txt 2>&1; tail -3 $O/pytest_conv.txt
TISE_CONV_DUO=0 timeout 300 python tools/split_layer_probe.py 500 > $O/layers_duo0.txt 2>&1
TISE_CONV_DUO=1 timeout 300 python tools/split_layer_probe.py 500 > $O/layers_duo1.txt 2>&1
head -2 $O/layers_duo0.txt | tail -1; head -2 $O/layers_duo1.txt | tail -1
for i in 1 2; do
for d in 0 1; do
TISE_CONV_DUO=$d timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-host-feed --no-kernel-probe --png-images 0 > $O/bench_duo${d}_$i.json 2> $O/bench_duo${d}_$i.err
python - <<PY
import json
j=[json.loads(l) for l in open("$O/bench_duo${d}_$i.json") if l.startswith("{")][-1]
print("DUO=$d run $i: value", round(j["value"]), "frac", round(j["roofline"]["frac"],4), "trunk ms", round(j["stage_ms_per_device_batch"]["trunk"],2), "fid", j["scores"]["fid"])
PY
done; done
