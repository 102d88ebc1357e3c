#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05l; mkdir -p $O
for e in 0 1; do echo "== TISE_CONV_EARLY=$e"; TISE_CONV_EARLY=$e timeout 300 python tools/conv_ablate.py fast 2>&1 | grep -v amdgpu | grep "6b1x1\|6a \|6e7x1\|5b1x1\|7c3x3" | cut -c1-140; done
for i in 1 2; do
for e in 0 1; do
TISE_CONV_EARLY=$e timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-host-feed --no-kernel-probe --png-images 0 > $O/bench_early${e}_$i.json 2> $O/bench_early${e}_$i.err
python - <<PY
import json
j=[json.loads(l) for l in open("$O/bench_early${e}_$i.json") if l.startswith("{")][-1]
print("EARLY=$e run $i: value", round(j["value"]), "frac", round(j["roofline"]["frac"],4), "trunk ms", round(j["stage_ms_per_device_batch"]["trunk"],2), "fid", j["scores"]["fid"])
PY
done; done
