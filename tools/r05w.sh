#!/bin/bash
# full GPU suite + smoke on the final commit of round 5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $O/gpu_suite.txt 2>&1; tail -20 $O/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
