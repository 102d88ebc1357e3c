#!/bin/bash
# round 5, first call: RCCL one-rank group + IPC probe, bias-free IS* head, driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
python tools/rccl_probe.py > $O/rccl_world1.txt 2> $O/rccl_world1.err
python -m pytest tests/test_gpu_rccl.py tests/test_gpu_pipeline.py -x -q -m gpu -k "rccl or ipc or is_ or fid_end_to_end or features_match or object_centric" > $O/pytest_sel.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
tail -5 $O/pytest_sel.txt
tail -c 1500 $O/rccl_world1.txt
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r05a/bench_driver.json").read().strip().splitlines()[-1])
print("value", j["value"], "collective", j["config"]["collective"], "allreduce_ms", j["allreduce_ms"], "ref", j["reference_side"], "fin", j["finalize_ms"], "parity", j["parity"], "frac", j["roofline"]["frac"])
PY
