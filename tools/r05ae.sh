#!/bin/bash
# final bench lines of round 5 on the last commit: the driver's command and the default run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05ae; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 600 $O/bench_driver_cmd.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
for f in ("bench_driver_cmd", "bench_default"):
    d = json.load(open(f"gpurun_out/r05ae/{f}.json"))
    print(f, round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d.get("png_feed", {}).get("value"), d["cpu_baseline"]["value"], d["config"]["collective"]["backend"])
PY
