#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
python tools/perclass_probe.py 80 300 > $O/perclass_probe.txt 2>&1
tail -3 $O/perclass_probe.txt
python -m pytest tests -x -q -m gpu --durations=25 > $O/gpu_suite.txt 2>&1
tail -40 $O/gpu_suite.txt
