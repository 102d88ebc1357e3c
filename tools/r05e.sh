#!/bin/bash
# round 5 evidence call: Winograd decision probe, per-layer roofline, new kernel tests, driver's bench command, rocprofv3 + PMC
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "classifier_layer" > $O/pytest_fc.txt 2>&1; tail -3 $O/pytest_fc.txt
python tools/winograd_probe.py 1500 > $O/winograd_probe.txt 2>&1; tail -4 $O/winograd_probe.txt
python tools/split_layer_probe.py 500 > $O/trunk_conv_layers.txt 2>&1; tail -3 $O/trunk_conv_layers.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python - <<'PY'
import json
j=[json.loads(l) for l in open("gpurun_out/r05e/bench_driver.json") if l.startswith("{")][-1]
print("value", j["value"], "frac", j["roofline"]["frac"], "stages", j["stage_ms_per_device_batch"], "fin", {k: round(v, 3) if isinstance(v, float) else v for k, v in j["finalize_ms"].items()})
print("png_feed", j["png_feed"]); print("host_feed", j["host_feed"]["images_per_s"], "parity", j["parity"]["dfid"], j["parity"]["dis"], "cc", j["cross_check"]["dfid"])
PY
