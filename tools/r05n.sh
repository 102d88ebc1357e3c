#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05n; mkdir -p $O
for i in 1 2 3 4; do
for e in 0 1; do
TISE_CONV_EARLY=$e timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-host-feed --no-kernel-probe --png-images 0 > $O/bench_early${e}_$i.json 2> $O/bench_early${e}_$i.err
python - <<PY
import json
j=[json.loads(l) for l in open("$O/bench_early${e}_$i.json") if l.startswith("{")][-1]
print("EARLY=$e run $i: value", round(j["value"]), "trunk ms", round(j["stage_ms_per_device_batch"]["trunk"],2))
PY
done; done
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python -m pytest tests -x -q -m gpu --durations=8 > $O/gpu_suite.txt 2>&1
tail -14 $O/gpu_suite.txt
