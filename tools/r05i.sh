#!/bin/bash
# round 5 final evidence: driver's command, default run, rocprofv3 kernel trace + six PMC passes (tools/run_profiles.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05i; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/run_profiles.sh r05i > $O/run_profiles.log 2>&1
python - <<'PY'
import json
for f in ("bench_driver", "bench_default"):
    j=[json.loads(l) for l in open(f"gpurun_out/r05i/{f}.json") if l.startswith("{")][-1]
    print(f, "value", round(j["value"]), "frac", round(j["roofline"]["frac"],4), "trunk", j["stage_ms_per_device_batch"], "sytrd", round(j["finalize_ms"]["sytrd"],2), "after-loop", round(j["finalize_ms"]["frechet_plus_is"],2), "allreduce", round(j["allreduce_ms"],3), "png", round(j["png_feed"]["images_per_s"]) if j.get("png_feed") else None, "traffic", j["roofline"]["traffic"])
PY
tail -12 $O/run_profiles.log
