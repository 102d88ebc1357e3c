#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05m; mkdir -p $O
for i in 1 2 3; do
for f in 0 0x4000; do
TISE_CONV_FLAGS=$f timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-host-feed --no-kernel-probe --png-images 0 > $O/bench_f${f}_$i.json 2> $O/bench_f${f}_$i.err
python - <<PY
import json
j=[json.loads(l) for l in open("$O/bench_f${f}_$i.json") if l.startswith("{")][-1]
print("FLAGS=$f run $i: value", round(j["value"]), "frac", round(j["roofline"]["frac"],4), "trunk ms", round(j["stage_ms_per_device_batch"]["trunk"],2), "fid", j["scores"]["fid"])
PY
done; done
