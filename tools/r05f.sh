#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=15 > $O/gpu_suite.txt 2>&1
tail -22 $O/gpu_suite.txt
